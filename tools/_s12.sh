mkdir -p gpurun_out/s12
timeout -k 10 400 python tools/c5_ab.py 11=2 > gpurun_out/s12/c5_ab11.txt 2>&1
grep "options\|chol" gpurun_out/s12/c5_ab11.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s12/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s12/tests.log
tail -3 gpurun_out/s12/tests.log
