"""Cuts the per-topic counter summaries out of kernel_pmc.json (scripts/pmc_kernels.sh):
    python scripts/cut_pmc.py <kernel_pmc.json> <out dir>
  gather_tile_pmc.json  the ray-grid gather kernel (what bench.py's `roofline.traffic` quotes: hbm_traffic_bytes_per_launch)
  modconv_pmc.json      every convolution / head kernel (mfma_busy_frac, instruction mix, LDS conflicts)
  render_pmc.json       the fused ray-marcher, sample_voxel and the density kernel
  fir_pmc.json          the FIR tile kernels (lds_bank_conflict_frac: VERDICT r3 item 2)"""
import json, os, sys

src, out = sys.argv[1], sys.argv[2]
d = json.load(open(src))
NOTE = ('rocprofv3 --pmc passes run separately with --kernel-trace only (scripts/pmc_kernels.sh over scripts/kernel_rooflines.py --eager); '
        'means over launches; *_SIZE in KB; full counter set in kernel_pmc.json')


def cut(pred):
    return {k: v for k, v in d.items() if pred(k)}


g = [k for k in d if 'triplane_sample_tile_pc_kernel' in k]
if g:
    rec = {'kernel': g[0].split('(')[0].replace('ide3d::(anonymous namespace)::', '').replace('void ', ''),
           'shape': 'N=4 images x 1 tri-plane, C=32, 256x256 channels_last, M=393216 samples/image, ray grid 64x64x96', 'note': NOTE}
    rec.update(d[g[0]])
    rec['hbm_traffic_bytes_per_launch'] = rec.get('hbm_traffic_bytes')
    json.dump(rec, open(os.path.join(out, 'gather_tile_pmc.json'), 'w'), indent=1)
json.dump({'note': NOTE, 'kernels': cut(lambda k: 'modconv' in k or 'head_split' in k or 'head_resident' in k)}, open(os.path.join(out, 'modconv_pmc.json'), 'w'), indent=1)
json.dump({'note': NOTE, 'kernels': cut(lambda k: 'render_rays' in k or 'sample_voxel' in k or 'density_kernel' in k)}, open(os.path.join(out, 'render_pmc.json'), 'w'), indent=1)
json.dump({'note': NOTE, 'kernels': cut(lambda k: 'upfirdn2d' in k)}, open(os.path.join(out, 'fir_pmc.json'), 'w'), indent=1)
print('cut', len(d), 'kernels')
