"""Generates tests/golden/ref_*.npz by RUNNING THE REFERENCE's own numpy code.

Run in the build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
The fixtures are data (seeded inputs + the reference's outputs); this script
and tools/ref_shim.py are the recipe that made them.  Reference functions
executed (paths relative to /root/reference):
    load_data.points_to_voxel                     load_data.py:695-771 (+ :593-641)
    load_data.generate_anchors                    load_data.py:1641-1685
    load_data.rbbox2d_to_near_bbox                load_data.py:535-547
    load_data.sparse_sum_for_anchors_mask         load_data.py:586-591
    load_data.fused_get_anchors_area              load_data.py:558-584
    load_data.center_to_corner_box2d              load_data.py:1525-1545
    load_data.corner_to_standup_nd_jit            load_data.py:1330-1340
    eval_helper_functions.second_box_decode       libraries/eval_helper_functions.py:388-461
    eval_helper_functions.nms_postprocess         libraries/eval_helper_functions.py:529-546
    eval_helper_functions.box_lidar_to_camera     libraries/eval_helper_functions.py:735-740
Not runnable here (TensorFlow): PillarFeatureNet, PointPillarsScatter, RPN, predict().
nms(), nms_gpu and nms_kernel run through the CUDA-model emulator of ref_shim.py: tools/gen_golden_kernels.py.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402

ld, ehf = ref_shim.load_reference()
import pp_amd  # noqa: E402  (only config + synth: host helpers, no GPU)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def edge_points(rng, n_feat=3):
    """Hand-placed cases: range boundaries, the z = 1.0 cell edge of the shipped
    config, duplicates, and > max_points points in one cell."""
    e = [
        [0.0, 0.0, 0.0], [6.40, 0.0, 0.0], [6.3999996, 0.0, 0.0], [-0.0, -2.56, -3.0],
        [1.0, 2.56, 0.0], [1.0, 2.5599999, 0.0], [1.0, -2.5600002, 0.0], [1.0, 0.0, 3.0],
        [1.0, 0.0, 2.9999998], [1.0, 0.0, 1.0], [1.0, 0.0, 0.99999994], [1.0, 0.0, -3.0000002],
        [0.08, 0.08, 0.5], [0.07999999, 0.08, 0.5], [0.16, 0.24, 0.5], [0.16000001, 0.24000001, 1.5],
        [3.2, 0.0, 0.0], [3.2, 0.0, 0.0], [3.2, 0.0, 0.0],  # duplicates
    ]
    e = np.array(e, dtype=np.float32)
    crowd = np.stack([2.0 + rng.uniform(0.001, 0.079, 70), 0.4 + rng.uniform(0.001, 0.079, 70),
                      rng.uniform(-0.5, 0.5, 70)], axis=1).astype(np.float32)
    crowd2 = np.stack([2.0 + rng.uniform(0.001, 0.079, 60), 0.4 + rng.uniform(0.001, 0.079, 60),
                       rng.uniform(1.1, 2.5, 60)], axis=1).astype(np.float32)  # same (y,x), second z cell
    pts = np.concatenate([e, crowd, crowd2], axis=0)
    if n_feat > 3:
        pts = np.concatenate([pts, rng.uniform(0, 1, (pts.shape[0], n_feat - 3)).astype(np.float32)], axis=1)
    return pts


def main():
    rng = np.random.default_rng(20261004)
    cfgA = pp_amd.config.pedestrian_d435i_config()
    dA = pp_amd.config.Derived(cfgA)
    cfgK = pp_amd.config.kitti_shaped_config()
    dK = pp_amd.config.Derived(cfgK)
    cfgT = pp_amd.config.tiny_config()
    dT = pp_amd.config.Derived(cfgT)

    # ---------------- a1 voxelise ----------------
    vox = {}

    def voxel_case(name, pts, d, max_points, max_voxels):
        v, c, n = ld.points_to_voxel(pts, d.voxel_size, d.pc_range, max_points, True, max_voxels)
        vox[name + "_points"] = pts
        vox[name + "_params"] = np.array([max_points, max_voxels], dtype=np.int64)
        vox[name + "_voxels"], vox[name + "_coors"], vox[name + "_num"] = v, c, n
        print(f"voxel case {name}: N={pts.shape[0]} -> P={v.shape[0]}, max num={n.max() if len(n) else 0}")
        return v, c, n

    p2k = np.concatenate([edge_points(rng), pp_amd.synth.d435i_cloud(0, 2000 - 149)], axis=0)
    p2k = p2k[rng.permutation(p2k.shape[0])]
    r2k = voxel_case("a2k", p2k, dA, 50, 12000)
    p16k = pp_amd.synth.d435i_cloud(1, 16384)
    r16k = voxel_case("a16k", p16k, dA, 50, 12000)
    voxel_case("brk", p2k, dA, 5, 300)             # pins the max_voxels break + small T truncation
    voxel_case("t100", p2k[:1500], dA, 100, 12000)   # BASELINE configs[0] alternates: T=100
    pk = np.concatenate([pp_amd.synth.kitti_cloud(0, 5000), edge_points(rng, 4) * np.float32([10, 15, 1, 1])], axis=0)
    voxel_case("kitti5k", pk, dK, 100, 12000)
    pt = np.concatenate([edge_points(rng)[:, :3] * np.float32([0.25, 0.25, 1.0]),
                         rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (400, 3)).astype(np.float32)], axis=0)
    voxel_case("tiny", pt, dT, 8, 500)
    np.savez_compressed(os.path.join(OUT, "ref_voxel.npz"), **vox)

    # ---------------- a3 anchors ----------------
    anc = {}
    for name, d in (("A", dA), ("T", dT), ("K", dK)):
        a = ld.generate_anchors(d.feature_map_size, d.anchor_cfg)["anchors"].reshape([-1, 7])
        bv = ld.rbbox2d_to_near_bbox(a[:, [0, 1, 3, 4, 6]])
        if name == "K":  # 3 MB: keep a digest + strided sample
            anc["K_sha256"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
            anc["K_bv_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(bv).tobytes()).digest(), dtype=np.uint8)
            anc["K_rows"] = a[::997].copy()
            anc["K_shape"] = np.array(a.shape)
        else:
            anc[name + "_anchors"] = a
            anc[name + "_bv"] = bv
    np.savez_compressed(os.path.join(OUT, "ref_anchors.npz"), **anc)

    # ---------------- a4 anchor mask ----------------
    msk = {}
    anchorsA = anc["A_anchors"]
    bvA = anc["A_bv"]
    for name, (v, c, n) in (("a2k", r2k), ("a16k", r16k)):
        dense = ld.sparse_sum_for_anchors_mask(c, tuple(dA.grid[::-1][1:]))
        dense = dense.cumsum(0).cumsum(1)
        area = ld.fused_get_anchors_area(dense, bvA, dA.voxel_size, dA.pc_range, dA.grid)
        msk[name + "_area"] = area
        msk[name + "_mask"] = (area > dA.anchor_area_threshold)
        print(f"mask case {name}: {int(msk[name + '_mask'].sum())} of {area.shape[0]} anchors kept")
    np.savez_compressed(os.path.join(OUT, "ref_mask.npz"), **msk)

    # ---------------- a9 decode, a8 corners / stand-up ----------------
    sel = rng.choice(anchorsA.shape[0], 300, replace=False)
    enc = (rng.standard_normal((300, 7)) * np.array([0.5, 0.5, 0.3, 0.2, 0.2, 0.2, 0.8])).astype(np.float32)
    dec = ehf.second_box_decode(enc, anchorsA[sel])
    bev = dec[..., [0, 1, 3, 4, 6]]
    corners = ld.center_to_corner_box2d(bev[:, :2], bev[:, 2:4], bev[:, 4])
    standup = ld.corner_to_standup_nd_jit(corners)
    np.savez_compressed(os.path.join(OUT, "ref_decode.npz"), enc=enc, anchors=anchorsA[sel], decoded=dec,
                        corners=corners, standup=standup)
    assert dec.dtype == np.float32 and standup.dtype == np.float32

    # ---------------- a11 host sweep ----------------
    nmsd = {}
    for n in (1, 37, 64, 100, 130):
        cb = -(-n // 64)
        m = np.zeros((n * cb,), dtype=np.uint64)
        for i in range(n):
            for b in range(cb):
                bits = rng.random(64) < 0.04
                word = 0
                for k in range(64):
                    j = b * 64 + k
                    if bits[k] and j > i and j < n:
                        word |= 1 << k
                m[i * cb + b] = np.uint64(word)
        keep = np.zeros((n,), dtype=np.int32)
        nk = ehf.nms_postprocess(keep, m, n)
        nmsd[f"n{n}_mask"] = m
        nmsd[f"n{n}_keep"] = keep[:nk].copy()
    np.savez_compressed(os.path.join(OUT, "ref_nms_post.npz"), **nmsd)

    # ---------------- a12 lidar -> camera ----------------
    rect, trv, _ = pp_amd.synth.default_calib()
    rect2 = rect.copy()
    rect2[:3, :3] = np.array([[0.9999, 0.0098, -0.0074], [-0.0099, 0.9999, -0.0043], [0.0074, 0.0044, 0.9999]], np.float32)
    trv2 = trv.copy()
    trv2[:3, 3] = [-0.004, -0.076, -0.272]
    cam1 = ehf.box_lidar_to_camera(dec[:50], rect, trv)
    cam2 = ehf.box_lidar_to_camera(dec[:50], rect2, trv2)
    assert cam1.dtype == np.float64
    np.savez_compressed(os.path.join(OUT, "ref_camera.npz"), boxes=dec[:50], rect=rect, trv=trv, cam=cam1,
                        rect2=rect2, trv2=trv2, cam2=cam2)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
