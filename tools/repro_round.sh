mkdir -p gpurun_out
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r5g_driver_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5g_driver_pytest.log
python3 -c 'import sys; sys.path.insert(0,"."); import __graft_entry__ as e; e.smoke()' > gpurun_out/r5g_driver_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r5g_driver_smoke.log
python bench.py > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err
python3 -m pytest tests/ -q -m gpu -p no:cacheprovider -s > gpurun_out/r5g_gpu_suite_s.log 2>&1; echo "rc=$?" >> gpurun_out/r5g_gpu_suite_s.log
tail -3 gpurun_out/r5g_driver_pytest.log; tail -3 gpurun_out/r5g_driver_smoke.log; cut -c1-200 gpurun_out/r5g_bench.json; tail -3 gpurun_out/r5g_gpu_suite_s.log
