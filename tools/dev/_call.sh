export PYTHONPATH=$PWD TMPDIR=/tmp
for i in 1 2; do
timeout 300 python tools/dev/clip_rate.py 2>&1 | grep "B=255"
CLEARCAM_CLIP_STAGE_INPUT=1 timeout 300 python tools/dev/clip_rate.py 2>&1 | grep "B=255" | sed 's/^/staged /'
done
