"""Write the network pickle the untouched reference drivers load (`--network <file>`): `dict(G_ema=G)` with a random-init
`training.triplane.TriPlaneGenerator` — what `legacy.load_network_pkl` (legacy.py:22-42) needs is `G_ema` being an nn.Module whose
class is importable (`training.triplane` is where viz/renderer.py:196 expects it).

    python scripts/make_random_init_pkl.py [--out random-init-ide3d-ffhq-64-512.pkl] [--seed 0] [--tiny]
    PYTHONPATH=ide-3d_amd:/path/to/IDE-3D python /path/to/IDE-3D/gen_images.py --network random-init-ide3d-ffhq-64-512.pkl --seeds 0-3 --outdir out
"""

import argparse
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd'))

import torch  # noqa: E402


def make(path, seed=0, tiny=False, tiny52=False):
    from training import triplane
    torch.manual_seed(seed)
    spec = None
    if tiny52:       # the tiny topology with the released model's decoder row: 32 colour features + 19 classes + sigma = 52 (extract_shapes.py:146)
        spec = triplane.tiny_spec(plane_channels=32, feature_channels=32, seg_channels=19)
    elif tiny:
        spec = triplane.tiny_spec()
    G = triplane.TriPlaneGenerator(spec).eval().requires_grad_(False)
    with open(path, 'wb') as f:
        pickle.dump(dict(G_ema=G, G=None, D=None, training_set_kwargs=None, augment_pipe=None), f)
    return G


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='random-init-ide3d-ffhq-64-512.pkl')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--tiny', action='store_true', help='the seconds-on-CPU test topology instead of the 64->512 generator')
    ap.add_argument('--tiny52', action='store_true', help='the tiny topology with a 52-wide decoder row (32 features + 19 classes + sigma)')
    a = ap.parse_args()
    G = make(a.out, a.seed, a.tiny, a.tiny52)
    print(f'wrote {a.out}: {sum(p.numel() for p in G.parameters()) / 1e6:.1f} M parameters, num_ws={G.num_ws}')
