for v in "" m16; do
  if [ -n "$v" ]; then export CLAMD_LIB=build/$v/libclamd.so; else unset CLAMD_LIB; fi
  echo "== variant: ${v:-default}"
  python tools/conv_ab.py bf16 0 2>&1 | grep -v amdgpu.ids
  CONV_MODE=dgrad python tools/conv_ab.py bf16 0 2>&1 | grep -v amdgpu.ids
  python tools/wgrad_ab.py bf16 0 2>&1 | grep -v amdgpu.ids | tail -12
done
