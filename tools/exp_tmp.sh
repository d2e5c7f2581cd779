set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests3.log 2>&1 || (tail -30 gpurun_out/r3_tests3.log; exit 1)
tail -2 gpurun_out/r3_tests3.log
tools/ab_headline.sh libvo_hip.so 2>&1 | tail -3
python tools/small_rate.py 127 1000
python tools/batch_frames.py 200
tools/prof_batch.sh r3d 200 2>&1 | grep -E "join|transform|pack|tri_|sum per"
