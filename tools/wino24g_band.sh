# Time and L2-miss bytes of the transform-free Winograd loop under every block order (clamd_tuning::wino_band), one layer shape per line of
# arguments:   bash tools/wino24g_band.sh "512 512 32" "1024 1024 16"      (where the re-read floor of V is)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' '_')
  python tools/wino24g_band.py $shape 10 | grep -v amdgpu
  rm -rf gpurun_out/band_$tag
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/band_$tag -o pmc -- python3 tools/wino24g_band.py $shape 3 > gpurun_out/band_$tag.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob('gpurun_out/band_$tag/**/pmc_counter_collection.csv', recursive=True)
rows = [r for fn in f for r in csv.DictReader(open(fn)) if r['Kernel_Name'].startswith('void clamd::wino24g_kernel') and r['Counter_Name'] == 'FETCH_SIZE']
rows.sort(key=lambda r: int(r['Dispatch_Id']))
bands = [b for b in (0, 1, 2, 4, 8, 16)]
vals = [float(r['Counter_Value']) * 1024 * 2 / 1e6 for r in rows]      # KiB, gfx950 half-count correction (MI355X_MICROARCH.md)
per = 4
print('FETCH_SIZE (MB per launch, x2 corrected), launches in band order:', [round(sum(vals[i:i + per]) / len(vals[i:i + per])) for i in range(0, len(vals), per)])
PY
done
