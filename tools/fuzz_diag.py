"""Diagnostic for one fuzz_parity seed: per frame, the detections of the HIP path and of the oracle side by side where they
differ (scores, the oracle's score gaps around the differing rows, IoUs of the involved boxes against the NMS threshold)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import pp_amd as pp
    import util_ref
    import fuzz_parity as fz
    from oracle import ref_numpy as rn
    seed = int(sys.argv[1])
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 32]))
    cfg = fz.random_config(pp, rng, B)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=8192)
    d = eng.d
    w = pp.weights.init_weights(d, seed=seed)
    eng.load_weights(w)
    frames = fz.random_frames(rng, d, B)
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    print("nms", d.nms_dict())
    for b in range(B):
        a = pp.VoxelNet._to_dict(dets[b], int(n[b]), b)
        r = ref["dets"][b]
        if r["scores"] is None or a["scores"] is None:
            continue
        if a["scores"].shape == r["scores"].shape and np.max(np.abs(a["box3d_lidar"] - r["box3d_lidar"])) < 1e-3:
            continue
        print(f"frame {b}: hip {a['scores'].shape[0]} oracle {r['scores'].shape[0]}")
        k = min(len(a["scores"]), len(r["scores"]))
        bad = [i for i in range(k) if np.max(np.abs(a["box3d_lidar"][i] - r["box3d_lidar"][i])) > 1e-3]
        print("  differing rows", bad[:20])
        i0 = bad[0]
        for i in range(max(0, i0 - 2), min(k, i0 + 4)):
            print(f"  row {i}: hip score {a['scores'][i]:.7f} box {np.round(a['box3d_lidar'][i], 4)} | oracle score {r['scores'][i]:.7f} box {np.round(r['box3d_lidar'][i], 4)}")
        # is the hip row i0 anywhere in the oracle list, and vice versa?
        da = np.max(np.abs(r["box3d_lidar"] - a["box3d_lidar"][i0]), axis=1)
        dr = np.max(np.abs(a["box3d_lidar"] - r["box3d_lidar"][i0]), axis=1)
        print(f"  hip row {i0} in oracle list at {int(np.argmin(da))} (dist {da.min():.2e}); oracle row {i0} in hip list at {int(np.argmin(dr))} (dist {dr.min():.2e})")
        # standup IoU (the +1 convention) of the two contested boxes against all higher-scored oracle boxes
        def standup(box):
            bev = box[None, [0, 1, 3, 4, 6]]
            c = rn.center_to_corner_box2d(bev[:, :2], bev[:, 2:4], bev[:, 4])
            return rn.corner_to_standup(c)[0]
        for tag, box in (("hip", a["box3d_lidar"][i0]), ("oracle", r["box3d_lidar"][i0])):
            sb = standup(box)
            ious = [rn.nms_iou(standup(r["box3d_lidar"][j]), sb) for j in range(i0)]
            close = [(j, round(float(v), 7)) for j, v in enumerate(ious) if abs(v - d.nms_iou_threshold) < 1e-3]
            print(f"  {tag} row {i0}: IoUs within 1e-3 of the threshold {d.nms_iou_threshold} against earlier oracle rows: {close}")
    eng.close()


if __name__ == "__main__":
    main()
