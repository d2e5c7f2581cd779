set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests5.log 2>&1 || (tail -40 gpurun_out/r3_tests5.log; exit 1)
tail -2 gpurun_out/r3_tests5.log
tools/ab_headline.sh libvo_hip.so 2>&1 | tail -3
python tools/small_rate.py 127 1000; python tools/small_rate.py 127 1000
python tools/batch_frames.py 200
