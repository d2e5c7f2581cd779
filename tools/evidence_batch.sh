set -e
TAG=r04
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log || (tail -20 gpurun_out/${TAG}_bench_stderr.log; exit 1)
echo bench done
tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
echo collect done
tools/pmc_frames.sh $TAG 200 > gpurun_out/${TAG}_sq_frames.txt
head -14 gpurun_out/${TAG}_sq_frames.txt
tools/pmc_match.sh $TAG > gpurun_out/${TAG}_pmc_match.txt 2>&1; head -14 gpurun_out/${TAG}_pmc_match.txt
