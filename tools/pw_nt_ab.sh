# K-chunks per staged K-step of the pointwise / ConvTranspose GEMM kernel (igemm.hip, Geo<MODE_PW>::NT): 2 (default, three
# workgroups per CU), 3 (two), 4 (one).  Measured on the four ConvTranspose shapes, fp32: 2 wins everywhere (occupancy beats
# fewer barriers: e.g. 128->64 @128^2 forward 182 / 220 / 256 us).      bash tools/pw_nt_ab.sh
set -e
for n in 2 3 4; do
  CLAMD_EXTRA_FLAGS="-DIGEMM_PW_NT=$n" python continual-learning_amd/build.py --force > /dev/null 2>&1
  echo "== IGEMM_PW_NT=$n"
  timeout -k 10 120 python tools/convt_ab.py fp32 0 2>&1 | grep -v amdgpu.ids
done
