"""Per-layer conv3x3 forward / data-gradient kernel times of the last full step in a rocprofv3 kernel trace (bs16, 256x256)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_kernel' in r['Kernel_Name']]
step = rows[idx[-3]:idx[-2]]
fwd = [(64,64,256),(64,128,128),(128,128,128),(128,256,64),(256,256,64),(256,512,32),(512,512,32),(512,1024,16),(1024,1024,16),(1024,512,32),(512,512,32),(512,256,64),(256,256,64),(256,128,128),(128,128,128),(128,64,256),(64,64,256)]
names = ['enc1.3','enc2.1','enc2.4','enc3.1','enc3.4','enc4.1','enc4.4','dec1.0','dec1.3','dec2.0','dec2.3','dec3.0','dec3.3','dec4.0','dec4.3','last.0','last.3']
ig = [r for r in step if ('igemm_kernel' in r['Kernel_Name'] and ', 0, 0, ' in r['Kernel_Name']) or 'igemm_ws_kernel' in r['Kernel_Name']]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
dg = dict(zip(reversed(names), ig[17:]))
tf = td = 0
for (n, (ci, co, hw), r) in zip(names, fwd, ig[:17]):
    fl = 2 * 16 * hw * hw * 9 * ci * co
    d = dg[n]
    k = lambda x: 'ws' if 'ws' in x['Kernel_Name'] else 'bs'
    tf += dur(r); td += dur(d)
    print(f'{n:8s} {ci:5d}->{co:5d} @{hw:3d} fwd[{k(r)}] {dur(r):7.1f} us {fl/dur(r)/1e6:7.1f} TF | dgrad[{k(d)}] {dur(d):7.1f} us {fl/dur(d)/1e6:7.1f} TF')
print(f'total fwd {tf/1e3:.3f} ms, dgrad {td/1e3:.3f} ms')
