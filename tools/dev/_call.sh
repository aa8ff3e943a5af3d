export PYTHONPATH=$PWD TMPDIR=/tmp
root=$PWD
for mp in 200000 100000 200000 100000; do CLEARCAM_STREAM_MIN_PIX=$mp timeout 300 python tools/dev/step_time.py f16h 2>&1 | grep "^f16h" | cut -c1-150 | sed "s/^/min_pix=$mp /"; done
CLEARCAM_STREAM_MIN_PIX=100000 CLEARCAM_PROFILE_CSV=$root/gpurun_out/r05u_minpix100k.csv timeout 300 python tools/dev/prof_csv.py 64 f16h 2>&1 | grep conv_ms | cut -c1-60
CLEARCAM_PROFILE_CSV=$root/gpurun_out/r05u_minpix200k.csv timeout 300 python tools/dev/prof_csv.py 64 f16h 2>&1 | grep conv_ms | cut -c1-60
