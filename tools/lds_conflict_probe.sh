# LDS bank conflicts of the conv3x3 kernel by counters: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS on tools/conv_ab.py (one
# deep layer), default build against a variant build (second argument of the loop below); per-kernel sums.   bash tools/lds_conflict_probe.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CONV_LAYERS='1024,512,32'
for v in default ${1:-diag}; do
  if [ $v != default ]; then export CLAMD_LIB=build/$v/libclamd.so; fi
  timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/ldsconf_$v -o pmc -- python3 tools/conv_ab.py bf16 0 > gpurun_out/ldsconf_$v.log 2>&1
  echo "rc=$?"
  python - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/ldsconf_$v/**/pmc_counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'igemm' in k: print('$v',k,dict(v))
PY
done
