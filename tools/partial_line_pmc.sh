# Counters behind "partial lines cost what full lines cost" (DESIGN section 4, round 4): the bf16 im2col, the loss kernel's bf16 NHWC copy and the filter
# pack, shipped kernels against a variant build of their predecessors
#   python continual-learning_amd/build.py --variant old3 -DIM2COL_FOUR_PIXELS -DCE_NO_EXCHANGE -DPACK_NO_RUN_PATH
# per kernel and launch: texture-addresser busy cycles, L1 -> L2 write requests, L2 -> memory write requests (all / 64-byte ones).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in shipped old3; do
  if [ $v != shipped ]; then export CLAMD_LIB=build/$v/libclamd.so; else unset CLAMD_LIB; fi
  for prog in boundary_time pack_time; do
    rm -rf gpurun_out/plpmc_${v}_$prog
    timeout -k 10 150 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/plpmc_${v}_$prog -o pmc -- python3 tools/$prog.py bf16 > gpurun_out/plpmc_${v}_$prog.log 2>&1 || echo "rocprofv3 failed ($v $prog)"
  done
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/plpmc_${v}_*/**/pmc_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].replace('void ', '').replace('clamd::', '').split('(')[0]
        if any(t in k for t in ('im2col', 'ce4_kernel', 'pack_kernel')):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(agg.items()):
    print('$v', k[:60].ljust(60), '  '.join(f'{n} {sum(x) / len(x):.3g}' for n, x in sorted(c.items())), f'({len(next(iter(c.values())))} launches)')
PY
done
