# A/B of the issue-order hint of the F(2x4) Winograd K loop (wino24.hip, W24_SCHED): where in the 48 MFMA slots of a chunk the
# LDS fragment reads and the transform VALU operations go.      bash tools/w24_sched_ab.sh
set -e
for n in 0 1 2 3 4; do
  CLAMD_EXTRA_FLAGS="-DW24_SCHED=$n" python continual-learning_amd/build.py --force > /dev/null 2>&1
  echo "== W24_SCHED=$n"
  timeout -k 10 200 python tools/wino24_ab.py 10 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4, $7, $8}' | tail -15
done
