# Timing ablation of the channels-in-the-lane epilogue of the persistent conv3x3 kernel (bf16): default build against variants built with
#   python continual-learning_amd/build.py --variant abl1 -DPWS_ABLATE_EPI=1      (no epilogue at all)
#   ... --variant abl2 -DPWS_ABLATE_EPI=2 (everything but the global stores)      ... --variant abl3 -DPWS_ABLATE_EPI=3 (no statistics)
# Results of the ablated builds are wrong by construction; only the times matter.   bash tools/pws_cl_ablate.sh [variants ...]
export CONV_LAYERS='64,64,256;128,64,256;128,128,128;256,256,64'
if [ $# -gt 0 ]; then V=("$@"); else V=("" abl1 abl2 abl3); fi
for v in "${V[@]}"; do
  if [ -n "$v" ]; then export CLAMD_LIB=build/$v/libclamd.so; else unset CLAMD_LIB; fi
  echo "== variant: ${v:-default}"
  python tools/conv_ab.py bf16 0 2>&1 | grep -v amdgpu.ids
  CONV_MODE=dgrad python tools/conv_ab.py bf16 0 2>&1 | grep -v amdgpu.ids
done
