export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
root=$PWD
CLEARCAM_PROFILE_CSV=$root/gpurun_out/r05g_yolo_per_launch.csv timeout 300 python tools/dev/prof_csv.py 64 f16h 2>&1 | grep -v amdgpu.ids | tail -2
CLEARCAM_STREAM=0 CLEARCAM_PROFILE_CSV=$root/gpurun_out/r05g_yolo_per_launch_nostream.csv timeout 300 python tools/dev/prof_csv.py 64 f16h 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/dev/step_time.py f16h,f16,f16s 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05g_step_time.txt
CLEARCAM_STREAM=0 timeout 300 python tools/dev/step_time.py f16h 2>&1 | grep -v amdgpu.ids | sed 's/^/nostream /' | tee -a gpurun_out/r05g_step_time.txt
