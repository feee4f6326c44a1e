"""Development check of h5lite against the real library on whatever HDF5 files a machine has.

    python tools/h5lite_crosscheck.py [dir ...]      # default: the PyTables test files of the image's Anaconda tree

Stage 1 (the image's /opt/conda/bin/python3.9, h5py on libhdf5) walks every file and records, per dataset and per
attribute, shape, kind and a digest of the values; stage 2 (this interpreter) does the same through h5lite.  An object
h5lite refuses (`Unsupported`) is counted, never an error; an object both read must agree bit for bit.  Not a test (the
files are not ours to ship and the other interpreter exists in the build image only): tests/test_h5lite.py carries the
committed fixtures; this tool is how the reader was exercised on files it was not written against.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CONDA_PY = "/opt/conda/bin/python3.9"
DEFAULT_DIRS = ["/opt/conda/lib/python3.9/site-packages/tables/tests",
                "/opt/conda/lib/python3.9/site-packages/tables/nodes/tests"]


def digest(a):
    """Canonical digest of an array of numbers / bytes / str (object arrays: the strings joined)."""
    a = np.asarray(a)
    if a.dtype.kind == "O":
        items = [x if isinstance(x, bytes) else str(x).encode("utf-8") for x in a.ravel().tolist()]
        return ["str", list(a.shape), hashlib.sha1(b"\0".join(items)).hexdigest()]
    if a.dtype.kind in "SU":
        items = [x if isinstance(x, bytes) else x.encode("utf-8") for x in a.ravel().tolist()]
        return ["str", list(a.shape), hashlib.sha1(b"\0".join(items)).hexdigest()]
    if a.dtype.kind in "iufb":
        b = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<")))
        return [a.dtype.kind + str(a.dtype.itemsize), list(a.shape), hashlib.sha1(b.tobytes()).hexdigest()]
    return ["other:" + str(a.dtype), list(a.shape), ""]


def walk_h5py(path):
    import h5py
    out = {}

    def attrs_of(node, name):
        try:
            for k in node.attrs:
                try:
                    out[f"{name}@{k}"] = digest(node.attrs[k])
                except Exception as ex:  # noqa: BLE001
                    out[f"{name}@{k}"] = ["error", type(ex).__name__]
        except Exception as ex:  # noqa: BLE001
            out[f"{name}@"] = ["error", type(ex).__name__]

    def rec(g, name):
        attrs_of(g, name)
        for k in g:
            link = g.get(k, getlink=True)
            if not isinstance(link, h5py.HardLink):
                continue
            n = g[k]
            child = name.rstrip("/") + "/" + k
            if isinstance(n, h5py.Group):
                rec(n, child)
            else:
                attrs_of(n, child)
                try:
                    out[child] = digest(n[()]) if n.shape is not None else ["null"]
                except Exception as ex:  # noqa: BLE001
                    out[child] = ["error", type(ex).__name__]
    with h5py.File(path, "r") as f:
        rec(f, "/")
    return out


def walk_h5lite(path):
    sys.path.insert(0, ROOT)
    import pp_amd as pp
    h5 = pp.h5lite
    out = {}

    def attrs_of(node, name):
        try:
            for k, v in node.attrs.items():
                out[f"{name}@{k}"] = digest(v)
        except h5.Unsupported as ex:
            out[f"{name}@"] = ["unsupported", str(ex)]

    def rec(g, name):
        attrs_of(g, name)
        try:
            keys = g.keys()
        except h5.Unsupported as ex:
            out[name + "/*"] = ["unsupported", str(ex)]
            return
        for k in keys:
            child = name.rstrip("/") + "/" + k
            try:
                n = g[k]
            except h5.Unsupported as ex:
                out[child] = ["unsupported", str(ex)]
                continue
            if isinstance(n, h5.Group):
                rec(n, child)
            else:
                attrs_of(n, child)
                try:
                    out[child] = digest(np.asarray(n)) if n.shape is not None else ["null"]
                except h5.Unsupported as ex:
                    out[child] = ["unsupported", str(ex)]
    try:
        with h5.File(path) as f:
            rec(f, "/")
    except h5.Unsupported as ex:
        out["/*"] = ["unsupported", str(ex)]
    return out


def main(argv):
    if argv and argv[0] == "--h5py":
        files = json.load(open(argv[1]))
        res = {}
        for p in files:
            try:
                res[p] = walk_h5py(p)
            except Exception as ex:  # noqa: BLE001
                res[p] = {"/*": ["error", type(ex).__name__]}
        json.dump(res, open(argv[2], "w"))
        return 0
    dirs = argv or DEFAULT_DIRS
    files = sorted(p for d in dirs for p in glob.glob(os.path.join(d, "**", "*.h5"), recursive=True))
    with tempfile.TemporaryDirectory() as td:
        lst, ref = os.path.join(td, "files.json"), os.path.join(td, "ref.json")
        json.dump(files, open(lst, "w"))
        subprocess.check_call([CONDA_PY, os.path.abspath(__file__), "--h5py", lst, ref])
        want = json.load(open(ref))
    stats = {"files": len(files), "objects": 0, "equal": 0, "unsupported": 0, "skipped_by_h5py": 0, "mismatch": 0}
    why = {}
    for p in files:
        got = walk_h5lite(p)
        dense = [v for k, v in got.items() if v[0] == "unsupported" and k.endswith("*")]
        for k, w in want[p].items():
            stats["objects"] += 1
            if w[0] in ("error", "other") or w[0].startswith("other"):
                stats["skipped_by_h5py"] += 1
                continue
            g = got.get(k)
            if g is None:
                # below a group (or in a file) h5lite refused as a whole, or behind an attribute set it refused
                parent_refused = dense or any(v[0] == "unsupported" and (k.startswith(kk.rstrip("*")) or kk.endswith("@") and
                                                                         k.startswith(kk)) for kk, v in got.items())
                if parent_refused:
                    stats["unsupported"] += 1
                    continue
                stats["mismatch"] += 1
                print("MISSING", p, k)
                continue
            if g[0] == "unsupported":
                stats["unsupported"] += 1
                why[g[1]] = why.get(g[1], 0) + 1
                continue
            if g == w:
                stats["equal"] += 1
            else:
                stats["mismatch"] += 1
                print("MISMATCH", p, k, w, g)
    print(json.dumps(stats))
    print("refused:", json.dumps(why, indent=1))
    return 1 if stats["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
