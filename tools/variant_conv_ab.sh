# conv_ab timings of the default build against build variants:   bash tools/variant_conv_ab.sh <dtype> <layers or ""> variant ...
dt=$1; shift
[ -n "$1" ] && export CONV_LAYERS="$1"; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export CLAMD_LIB=build/$v/libclamd.so; else unset CLAMD_LIB; fi
  echo "== variant: ${v:-default}"
  python tools/conv_ab.py $dt 0 2>&1 | grep -v amdgpu.ids
  CONV_MODE=dgrad python tools/conv_ab.py $dt 0 2>&1 | grep -v amdgpu.ids
done
