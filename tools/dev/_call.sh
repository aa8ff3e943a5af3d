export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
for lv in 0 1 2; do CLEARCAM_FUSE_CSP=$lv timeout 600 python tools/dev/step_time.py f16h,f16 2>&1 | grep -v amdgpu.ids | sed "s/^/fuse_csp=$lv /"; done | tee gpurun_out/r05i_fuse_csp_levels.txt
