"""Randomised whole-path parity soak on the GPU box: configurations, batch sizes and clouds the tests do not pin.

    python tools/fuzz_parity.py [--seconds 300] [--seed0 1000] [--out gpurun_out/fuzz.log]

Per case: a random reference-schema configuration (grid, first stride, z cells, point features, channel widths, layer
counts, class count, direction head, distance feature, pillar caps, NMS sizes and thresholds), a random batch size from
both sides of the engine's kernel selection (a handful of frames: the split-K small-map kernels; dozens: the
persistent ones), random clouds (empty frames, points outside the range, crowded pillars).  The HIP path through the
C-ABI is compared with the oracle exactly as tests/test_gpu_parity.py::test_random_small_configs_end_to_end does
(pillars / anchor mask bit-exact, head maps within 1e-4, boxes within 1e-4 + 1e-4 of their size; rows that
changed places must share their score: the reference's order of equal scores is implementation-defined).  TEST INFRASTRUCTURE: imports oracle/.
Prints one line per case and a summary; exit code 1 on any mismatch (the failing seed reproduces the case).
"""
import argparse
import copy
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TOL = 1e-4


LARGE = False       # --large: KITTI-sized, mostly empty grids (the sparse first layer, the pillar-centric PFN, frame sub-ranges)


def random_config(pp, rng, B):
    cfg = copy.deepcopy(pp.config.pedestrian_d435i_config(B))
    s1 = int(rng.choice([1, 2]))
    nx, ny = 4 * s1 * int(rng.integers(3, 14)), 4 * s1 * int(rng.integers(2, 12))
    if LARGE:
        nx, ny = 8 * int(rng.integers(20, 63)), 8 * int(rng.integers(20, 63))
    v = float(rng.choice([0.08, 0.16]))
    nz2 = bool(rng.integers(0, 2))
    zr = (-3.0, 3.0) if nz2 else (-3.0, 1.0)
    x0, y0 = 0.0, -ny * v / 2
    F = int(rng.choice([3, 4]))
    C = int(rng.choice([32, 64, 128]))
    filters = [int(rng.choice([32, 64, 128])), int(rng.choice([32, 64, 128])), int(rng.choice([64, 128, 256]))]
    up = int(rng.choice([32, 64, 128]))
    ncls = int(rng.choice([1, 1, 2, 3]))
    cfg["eval_input_reader"].update(batch_size=B, feature_map_size=[1, ny // s1, nx // s1], num_point_features=F)
    if ncls > 1:
        cfg["eval_input_reader"]["desired_objects"] = ["Pedestrian", "Cyclist", "Car"][:ncls]
    s = cfg["model"]["second"]
    s["num_point_features"] = F
    s["num_class"] = ncls
    s["use_direction_classifier"] = bool(rng.integers(0, 4) > 0)
    s["voxel_generator"].update(point_cloud_range=[x0, y0, zr[0], x0 + nx * v, y0 + ny * v, zr[1]],
                                voxel_size=[v, v, 4.0], max_number_of_points_per_voxel=int(rng.choice([5, 12, 50, 100])),
                                max_number_of_voxels=int(rng.choice([150, 2000, 12000])))
    s["voxel_feature_extractor"]["num_filters"] = C
    s["voxel_feature_extractor"]["with_distance"] = bool(rng.integers(0, 3) == 0)
    s["rpn"].update(layer_nums=[int(rng.integers(1, 4)) for _ in range(3)], layer_strides=[s1, 2, 2],
                    num_filters=filters, upsample_strides=[1, 2, 4], num_upsample_filters=[up] * 3)
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(
        strides=[v * s1, v * s1, 0.0], offsets=[x0 + v * s1, y0, -1.465])
    s["nms_pre_max_size"] = int(rng.choice([50, 300, 1000]))
    s["nms_post_max_size"] = int(rng.choice([5, 100, 300]))
    s["nms_score_threshold"] = float(rng.choice([0.05, 0.3, 0.5]))
    s["nms_iou_threshold"] = float(rng.choice([0.1, 0.5, 0.7]))
    return cfg


def random_frames(rng, d, B):
    lo, hi = np.array(d.pc_range[:3]), np.array(d.pc_range[3:])
    frames = []
    for b in range(B):
        kind = int(rng.integers(0, 8))
        n = 0 if kind == 0 else int(rng.integers(1, 60)) if kind == 1 else int(rng.integers(300, 6000))
        if LARGE and kind > 1:
            n = int(rng.integers(2000, 20000))
        xyz = rng.uniform(lo - 0.2, hi + 0.2, (n, 3))
        if kind == 2 and n:                      # a crowd in a few pillars
            c = rng.uniform(lo, hi, (4, 3))
            xyz[: n // 2] = c[rng.integers(0, 4, n // 2)] + rng.uniform(-0.03, 0.03, (n // 2, 3))
        extra = rng.uniform(0, 1, (n, d.num_point_features - 3))
        frames.append(np.concatenate([xyz, extra], axis=1).astype(np.float32))
    return frames


def one_case(pp, util_ref, seed):
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 32])) if not LARGE else int(rng.choice([1, 2, 4, 9]))
    cfg = random_config(pp, rng, B)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=8192 if not LARGE else 20480)
    try:
        d = eng.d
        w = pp.weights.init_weights(d, seed=seed)
        eng.load_weights(w)
        frames = random_frames(rng, d, B)
        rect, trv, p2 = pp.synth.default_calib()
        ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
        dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
        im = eng.intermediates()
        for b in range(B):
            fr = ref["frames"][b]
            P = fr["coordinates"].shape[0]
            assert im["n_pillars"][b] == P, f"frame {b}: {im['n_pillars'][b]} pillars, oracle {P}"
            assert np.array_equal(im["coors"][b, :P], fr["coordinates"]), f"frame {b}: coordinates"
            assert np.array_equal(im["num_points"][b, :P], fr["num_points"]), f"frame {b}: num_points"
            assert np.array_equal(im["anchors_mask"][b].astype(bool), fr["anchors_mask"]), f"frame {b}: anchor mask"
        worst = 0.0
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            if ref["preds"].get(k) is None:
                continue
            err = float(np.max(np.abs(im[k] - ref["preds"][k]))) if im[k].size else 0.0
            worst = max(worst, err)
            assert err <= TOL, f"{k}: max abs err {err:.3g}"
        ndet, ties = 0, 0
        for b in range(B):
            a = pp.VoxelNet._to_dict(dets[b], int(n[b]), b)
            r = ref["dets"][b]
            if r["scores"] is None:
                assert a["scores"] is None, f"frame {b}: {int(n[b])} detections, oracle none"
                continue
            assert a["scores"] is not None and a["scores"].shape == r["scores"].shape, \
                f"frame {b}: {int(n[b])} detections, oracle {r['scores'].shape[0]}"
            k = r["scores"].shape[0]
            ndet += k
            rows_a = np.concatenate([a["box3d_lidar"], a["box3d_camera"], a["scores"][:, None], a["label_preds"][:, None]], axis=1)
            rows_r = np.concatenate([r["box3d_lidar"], r["box3d_camera"], r["scores"][:, None], r["label_preds"][:, None]], axis=1)
            # decoded sizes are exp(t) * anchor: an error e of the head value t (bar: 1e-4) is a RELATIVE error e of the size,
            # and random weights produce t = 17 (a 2e7 m box, whose t carries 1e-5 of float32 round-off) -- rows are
            # compared to 1e-4 + 1e-4 * |oracle| (in units of that bound)
            scale = lambda ref_rows: TOL + 1e-4 * np.abs(ref_rows)      # noqa: E731
            if np.max(np.abs(rows_a - rows_r) / scale(rows_r)) <= 1.0:
                continue
            # Not the same rows in the same order.  The one accepted reason (deviation 2): the order of
            # EQUAL scores is implementation-defined in the reference (np.argpartition / argsort) -- the same boxes must
            # then be there, and a box may only have moved past boxes whose score it shares to within twice this case's
            # measured head-map error (a logit error e moves a score by at most e / 4).
            dist = np.max(np.abs(rows_a[:, None, :] - rows_r[None, :, :]) / scale(rows_r)[None, :, :], axis=2) * TOL
            tie = max(2e-6, 2 * worst)
            sc = r["scores"]
            # rows of one list without a partner in the other: accepted only at the CUT -- the selection keeps the best
            # candidates (the reference: top-100 by np.argpartition; the kernel: ties broken by the lower anchor index), and
            # when the last place is shared by equal scores the two implementations may keep different anchors.  Such rows
            # must carry the lowest score of their list, and the same one (to the tie bound) in both lists.
            a_un = [i for i in range(k) if dist[i].min() > TOL]
            r_un = [j for j in range(k) if dist[:, j].min() > TOL]
            if a_un or r_un:
                smin = float(sc.min())
                at_cut = all(abs(float(a["scores"][i]) - smin) <= tie for i in a_un) and all(abs(float(sc[j]) - smin) <= tie for j in r_un)
                if not (at_cut and len(a_un) == len(r_un)):
                    i = a_un[0] if a_un else int(np.argmin(dist[:, r_un[0]]))
                    raise AssertionError(f"frame {b}: different boxes ({len(a_un)} hip rows / {len(r_un)} oracle rows without a partner; "
                                         f"hip row {i}: {np.array2string(rows_a[i], precision=6)}; lowest oracle score {smin:.7f})")
            keep_a = [i for i in range(k) if i not in a_un]
            perm = {i: int(np.argmin(dist[i])) for i in keep_a}
            assert len(set(perm.values())) == len(keep_a), f"frame {b}: two hip rows match one oracle row"
            for i, j in perm.items():
                lo_, hi_ = min(i, j), max(i, j)
                assert float(np.max(sc[lo_:hi_ + 1]) - np.min(sc[lo_:hi_ + 1])) <= tie, \
                    f"frame {b}: box {i} moved to {j} across scores {sc[lo_:hi_ + 1]}"
            ties += 1
        return f"B={B} grid={d.grid[0]}x{d.grid[1]}x{d.grid[2]} C={d.pfn_filters} f={d.num_filters} L={d.layer_nums} " \
               f"cls={d.num_class} dir={int(d.use_direction_classifier)} dets={ndet} maxerr={worst:.2e}" + (f" tie-order-frames={ties}" if ties else "")
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--seeds", default=None, help="comma-separated seeds to run instead of the timed sweep")
    ap.add_argument("--large", action="store_true", help="KITTI-sized grids (160..496 cells a side), up to 20 000 points, batches 1..9")
    a = ap.parse_args()
    global LARGE
    LARGE = a.large
    import pp_amd as pp
    import util_ref
    pp._lib.build()
    pp._lib.lib()
    out = open(a.out, "w") if a.out else None

    def say(line):
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()
    t0, seed, bad, n = time.time(), a.seed0, [], 0
    todo = [int(v) for v in a.seeds.split(",")] if a.seeds else None
    while (todo is None and time.time() - t0 < a.seconds and n < a.max_cases) or (todo is not None and n < len(todo)):
        if todo is not None:
            seed = todo[n]
        try:
            say(f"seed {seed}: ok  {one_case(pp, util_ref, seed)}")
        except AssertionError as ex:
            bad.append(seed)
            say(f"seed {seed}: MISMATCH " + " ".join(str(ex).split())[:900])
        except Exception as ex:  # noqa: BLE001
            bad.append(seed)
            say(f"seed {seed}: ERROR {type(ex).__name__}: {str(ex)[:300]}")
            say(traceback.format_exc()[-1500:])
        seed += 1
        n += 1
    say(f"{n} cases in {time.time() - t0:.0f} s, {len(bad)} bad: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
