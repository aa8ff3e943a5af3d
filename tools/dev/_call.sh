export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05z_pytest_gpu.txt
timeout 900 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-260 | tee gpurun_out/r05z_smoke.txt
timeout 2400 bash tools/measure_round.sh r05z > gpurun_out/r05z_measure.log 2>&1; echo "measure rc=$?"
tail -c 600 gpurun_out/r05z_bench_line.json
