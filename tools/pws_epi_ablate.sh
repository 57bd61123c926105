# Timing ablation of the persistent conv3x3 kernel's epilogue (igemm_pws.hip).  Builds: default; 1 = no epilogue; 2 = no
# global stores.  Results of the ablated builds are wrong by construction; only the times matter.
#   bash tools/pws_epi_ablate.sh [flags ...]      e.g.  bash tools/pws_epi_ablate.sh -DPWS_ABLATE_EPI=2
set -e
export CONV_LAYERS='64,64,256;128,64,256;128,128,128;256,128,128;256,256,64'
if [ $# -gt 0 ]; then FL=("$@"); else FL=("" "-DPWS_ABLATE_EPI=1" "-DPWS_ABLATE_EPI=2"); fi
for f in "${FL[@]}"; do
  CLAMD_EXTRA_FLAGS="$f" python continual-learning_amd/build.py --force > /dev/null 2>&1
  echo "== flags: $f"
  timeout -k 10 120 python tools/conv_ab.py bf16 0 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/conv_ab.py bf16x3 0 2>&1 | grep -v amdgpu.ids
done
