"""Density-cube extraction: the `extract_shapes.py` query loop (extract_shapes.py:74-150) on the HIP ray-marcher ops.

The reference builds a `voxel_resolution`^3 lattice on the host, scales it by 0.9, evaluates the generator's tri-planes
once and then queries `G.synthesis.renderer.sample_voxel` in chunks of `max_batch` points, keeping the density column.
Here each chunk is ONE launch of the fused gather + decoder kernel in its density-only mode with the lattice points
generated in registers (`ide3d_density_lattice`: no 201 MB coordinate array, 4 B written per point instead of 208 B),
using the reference's fp32 arithmetic operation by operation (including its float-division quirk: the y and x
"indices" are not floored, extract_shapes.py:84-86).  With 288 GB of HBM the whole 256^3 cube is one launch
(`max_batch=None`).

`save_density_cube` writes the two files of extract_shapes.py:191-194 — `<name>.npy` and an MRC2014 volume
(`mrcfile.new_mmap(..., mrc_mode=2)` there; here a dependency-free writer of the same 1024-byte header + float32
voxels, since `mrcfile` is not installed).  Marching cubes / .ply export (extract_shapes.py:27-70,195-219) is a further
host-side consumer of the cube: out of scope.
"""

import os
import struct

import numpy as np
import torch


def lattice_points(n, voxel_size, corner, scale, first, count, device='cpu'):
    """Points [first, first + count) of the lattice, already scaled -> [count, 3] float32.  On a CUDA device this is the
    HIP kernel `ide3d_lattice_points` (IEEE division / fmod, one rounding per operation: bit-equal to the host code
    below — torch's own GPU division by a scalar multiplies by the reciprocal and is NOT)."""
    device = torch.device(device)
    if device.type == 'cuda':
        from training import volumetric_rendering as vr
        vr._init()
        return vr._plugin.lattice_points(n, voxel_size, corner, scale, first, count, device)
    idx = torch.arange(first, first + count, 1, dtype=torch.int64)
    fidx = idx.float()
    s2 = (idx % n).float()
    s1 = (fidx / n) % n
    s0 = ((fidx / n) / n) % n
    pts = torch.stack([s0 * voxel_size + float(corner[2]), s1 * voxel_size + float(corner[1]), s2 * voxel_size + float(corner[0])], dim=1)
    return pts if scale == 1 else scale * pts


def create_samples(N=512, voxel_origin=(0, 0, 0), cube_length=2.0, device='cpu'):
    """Lattice of extract_shapes.py:74-96 -> (samples [1, N^3, 3] float32, voxel_origin (corner) float64 [3], voxel_size).

    Column 2 is the fastest index; columns 1 and 0 are `(i / N) % N` and `((i / N) / N) % N` evaluated in fp32 *without*
    flooring, exactly as the reference does (so points are sheared by a sub-voxel amount — kept for parity)."""
    voxel_origin = np.array(voxel_origin) - cube_length / 2
    voxel_size = cube_length / (N - 1)
    samples = lattice_points(N, voxel_size, voxel_origin, 1.0, 0, N ** 3, device)
    return samples.unsqueeze(0), voxel_origin, voxel_size


def split_ws(synthesis, ws):
    """extract_shapes.py:113-127: per-block slices of ws (`num_conv + num_torgb` wide, advancing by `num_conv`)."""
    assert ws.ndim == 3 and ws.shape[1] == synthesis.num_ws and ws.shape[2] == synthesis.w_dim
    ws = ws.to(torch.float32)
    voxel_ws, block_ws, w_idx = [], [], 0
    for res in synthesis.voxel_block_resolutions:
        block = getattr(synthesis, f'vb{res}')
        voxel_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
        w_idx += block.num_conv
    for res in synthesis.block_resolutions:
        block = getattr(synthesis, f'b{res}')
        block_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
        w_idx += block.num_conv
    return voxel_ws, block_ws


def triplanes_from_ws(synthesis, ws, noise_mode='random'):
    """extract_shapes.py:129-132: run the `vb*` blocks -> (img_v, seg_v).  The reference calls the blocks without a
    noise_mode, i.e. with the layers' default 'random' per-pixel noise; pass 'const' / 'none' for reproducible cubes."""
    voxel_ws, _ = split_ws(synthesis, ws)
    x_v = img_v = seg_v = None
    for res, cur_ws in zip(synthesis.voxel_block_resolutions, voxel_ws):
        x_v, img_v, seg_v = getattr(synthesis, f'vb{res}')(x_v, img_v, cur_ws, condition_img=seg_v, noise_mode=noise_mode)
    return img_v, seg_v


def density_cube(renderer, img_v, seg_v, voxel_resolution=256, max_batch=100000, voxel_origin=(0, 0, 0), cube_length=2.0,
                 scale=0.9, materialize=False):
    """The chunked query loop (extract_shapes.py:103-105,144-149) -> sigma [B, N, N, N] float32 on the planes' device.
    `max_batch=None` queries the whole lattice in one launch.  By default the points of a chunk are generated inside the
    density kernel (`renderer.density_lattice`); `materialize=True` builds the [1, N^3, 3] array first like the reference."""
    device = img_v.device
    b, total = img_v.shape[0], voxel_resolution ** 3
    corner = np.array(voxel_origin) - cube_length / 2
    voxel_size = cube_length / (voxel_resolution - 1)
    samples = None
    if materialize:
        samples = scale * create_samples(voxel_resolution, voxel_origin, cube_length, device=device)[0]
        if b > 1:
            samples = samples.expand(b, -1, -1)
    sigmas = torch.zeros((b, total), device=device)
    step = total if not max_batch else int(max_batch)
    head = 0
    while head < total:
        cnt = min(step, total - head)
        if materialize:
            sig = renderer.sample_voxel(img_v, seg_v, samples[:, head:head + cnt], sigma_only=True)
        else:
            sig = renderer.density_lattice(img_v, seg_v, voxel_resolution, voxel_size, corner, scale, head, cnt)
        sigmas[:, head:head + cnt] = sig.reshape(b, -1)
        head += cnt
    return sigmas.reshape(b, voxel_resolution, voxel_resolution, voxel_resolution)


def sample_generator_ide3d(generator, aux_img_net, z, c, max_batch=100000, voxel_resolution=256, voxel_origin=(0, 0, 0),
                           cube_length=2.0, psi=0.5, to_numpy=True, noise_mode='random', **kwargs):
    """extract_shapes.py:99-150 (same positional arguments; `aux_img_net` is unused there too).  Returns the density cube
    [N, N, N] — a numpy array like the reference, or the device tensor with `to_numpy=False`."""
    with torch.no_grad():
        ws = generator.mapping(z, c, truncation_psi=psi)
        img_v, seg_v = triplanes_from_ws(generator.synthesis, ws, noise_mode=noise_mode)
        sig = density_cube(generator.synthesis.renderer, img_v, seg_v, voxel_resolution=voxel_resolution, max_batch=max_batch,
                           voxel_origin=voxel_origin, cube_length=cube_length)
    sig = sig.reshape(voxel_resolution, voxel_resolution, voxel_resolution)
    return sig.cpu().numpy() if to_numpy else sig


def write_mrc(path, volume: np.ndarray):
    """MRC2014 file, mode 2 (float32), axis order (sections, rows, columns) = volume.shape like `mrcfile` writes it:
    nx = shape[2] (fastest), ny = shape[1], nz = shape[0]; cell 0 / sampling equal to the grid (what mrcfile.new_mmap
    leaves when only `data` is assigned, extract_shapes.py:191-192); header statistics filled in."""
    v = np.ascontiguousarray(volume, dtype=np.float32)
    assert v.ndim == 3
    nz, ny, nx = v.shape
    h = bytearray(1024)
    struct.pack_into('<3i', h, 0, nx, ny, nz)                 # NX NY NZ
    struct.pack_into('<i', h, 12, 2)                          # MODE 2 = float32
    struct.pack_into('<3i', h, 16, 0, 0, 0)                   # NXSTART NYSTART NZSTART
    struct.pack_into('<3i', h, 28, nx, ny, nz)                # MX MY MZ
    struct.pack_into('<3f', h, 40, 0.0, 0.0, 0.0)             # CELLA (unset)
    struct.pack_into('<3f', h, 52, 90.0, 90.0, 90.0)          # CELLB
    struct.pack_into('<3i', h, 64, 1, 2, 3)                   # MAPC MAPR MAPS
    struct.pack_into('<3f', h, 76, float(v.min()), float(v.max()), float(v.mean(dtype=np.float64)))   # DMIN DMAX DMEAN
    struct.pack_into('<i', h, 88, 1)                          # ISPG: 1 = volume
    struct.pack_into('<i', h, 92, 0)                          # NSYMBT
    h[104:108] = b'\x00\x00\x00\x00'                          # EXTTYP
    struct.pack_into('<i', h, 108, 20140)                     # NVERSION
    h[208:212] = b'MAP '
    h[212:216] = bytes([0x44, 0x44, 0x00, 0x00])              # little-endian machine stamp
    struct.pack_into('<f', h, 216, float(v.std(dtype=np.float64)))   # RMS
    struct.pack_into('<i', h, 220, 0)                         # NLABL
    with open(path, 'wb') as f:
        f.write(bytes(h))
        f.write(v.tobytes())


def read_mrc(path) -> np.ndarray:
    """Inverse of `write_mrc` for mode-2 files without extended header (used by the tests)."""
    with open(path, 'rb') as f:
        h = f.read(1024)
        nx, ny, nz, mode = struct.unpack_from('<4i', h, 0)
        assert mode == 2 and h[208:212] == b'MAP ' and struct.unpack_from('<i', h, 92)[0] == 0
        return np.frombuffer(f.read(), dtype='<f4').reshape(nz, ny, nx)


def save_density_cube(outdir, name, voxel_grid: np.ndarray):
    """extract_shapes.py:191-194: `<name>.mrc` + `<name>.npy`."""
    os.makedirs(outdir, exist_ok=True)
    write_mrc(os.path.join(outdir, f'{name}.mrc'), voxel_grid)
    np.save(os.path.join(outdir, f'{name}.npy'), voxel_grid)
