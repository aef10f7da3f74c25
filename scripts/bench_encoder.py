"""the "edit -> re-render" loop on one GPU (Painter/run_UI.py:193-199).
face_parsing (BiSeNet, 512x512) -> HybridEncoder(512, 10 appearance + 8 geometry latents) on the image + the 19-channel one-hot label map ->
G.synthesis of the predicted ws.  Prints one JSON line with the latencies (batch 1) and the encoder / parser throughput (batch 4)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from training import encoders, face_parsing, triplane
from torch_utils import hip_plugin

dev = torch.device('cuda:0')
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
E = encoders.HybridEncoder(G.img_resolution, 10, 8, G.w_dim).eval().to(dev)
P = face_parsing.BiSeNet(n_classes=20).eval().requires_grad_(False).to(dev)
cam = triplane.camera_label(0.2, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {}
with torch.no_grad():
    for b in (1, 4):
        img = torch.randn(b, 3, 512, 512, device=dev).clamp(-1, 1); seg = torch.randn(b, 19, 512, 512, device=dev)
        ms = timed(lambda: E(img, seg))
        res[f'encoder_b{b}'] = dict(ms=ms, images_per_s=b / ms * 1e3)
        ms = timed(lambda: face_parsing.face_parsing(img, P))
        res[f'face_parsing_b{b}'] = dict(ms=ms, images_per_s=b / ms * 1e3)
    img, seg = img[:1], seg[:1]
    ws = E(img, seg) * 0.05 + G.mapping.w_avg                      # keep the random-init latents in a sane range
    ms = timed(lambda: G.synthesis(E(img, seg) * 0.05 + G.mapping.w_avg, c=cam, noise_mode='const', return_seg=True))
    res['edit_to_rerender_b1'] = dict(ms=ms)
    # the whole loop of Painter/run_UI.py:193-199: render the current latents, parse the render, encode (image, labels), re-render
    def edit_loop():
        cur = G.synthesis(ws, c=cam, noise_mode='const')
        lab = face_parsing.face_parsing(cur, P)
        rec = E(cur, lab) * 0.05 + G.mapping.w_avg
        return G.synthesis(rec, c=cam, noise_mode='const', return_seg=True)
    res['render_parse_encode_rerender_b1'] = dict(ms=timed(edit_loop))
    res['parse_encode_rerender_b1'] = dict(ms=timed(lambda: G.synthesis(E(img, face_parsing.face_parsing(img, P)) * 0.05 + G.mapping.w_avg, c=cam, noise_mode='const', return_seg=True)))
print(json.dumps(dict(metric='HybridEncoder forward + re-render', **res, native_launches=dict(hip_plugin.CALLS))))
