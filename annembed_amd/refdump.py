"""Reader (and, for self-checks, writer) of the raw dump a real annembed run leaves behind through the `ref_dump` test of INTEGRATION.md
section 8: ten little-endian files `ref_*` next to each other.  The reference pin of the stages that have no numeric test upstream
(to_proba_edges, embedded scales, cross-entropy value, initial / final embedding)."""
import os

import numpy as np

FILES = {"indptr.u64": np.uint64, "nbr.u32": np.uint32, "dist.f32": np.float32, "proba.f32": np.float32, "scale.f32": np.float32,
         "y0.f32": np.float32, "y.f32": np.float32, "emb_scale.f32": np.float32, "ce.f64": np.float64}


def available(directory):
    return all(os.path.exists(os.path.join(directory, "ref_" + f)) for f in list(FILES) + ["meta.txt"])


def read(directory):
    """-> dict: meta (str -> float), indptr, nbr, dist, proba, scale, y0 [n, d], y [n, d], emb_scale, ce [2]"""
    out = {"meta": {}}
    with open(os.path.join(directory, "ref_meta.txt")) as f:
        for line in f:
            parts = line.split()
            if len(parts) == 2:
                out["meta"][parts[0]] = float(parts[1])
    for name, dt in FILES.items():
        out[name.split(".")[0]] = np.fromfile(os.path.join(directory, "ref_" + name), dtype=np.dtype(dt).newbyteorder("<")).astype(dt)
    n, d = int(out["meta"]["n"]), int(out["meta"]["asked_dim"])
    if len(out["indptr"]) != n + 1 or int(out["indptr"][-1]) != len(out["nbr"]) or len(out["dist"]) != len(out["nbr"]) or len(out["proba"]) != len(out["nbr"]):
        raise ValueError("reference dump: the graph arrays do not fit together")
    if len(out["scale"]) != n or len(out["emb_scale"]) != n or len(out["y0"]) != n * d or len(out["y"]) != n * d or len(out["ce"]) != 2:
        raise ValueError("reference dump: the per-node arrays do not fit n = %d, asked_dim = %d" % (n, d))
    out["y0"] = out["y0"].reshape(n, d)
    out["y"] = out["y"].reshape(n, d)
    return out


def write(directory, indptr, nbr, dist, proba, scale, y0, y, emb_scale, ce, meta):
    """the same format (tests: a dump made by the oracle keeps the reader and the checks alive where no reference dump exists)"""
    os.makedirs(directory, exist_ok=True)
    arrays = {"indptr.u64": indptr, "nbr.u32": nbr, "dist.f32": dist, "proba.f32": proba, "scale.f32": scale, "y0.f32": y0, "y.f32": y,
              "emb_scale.f32": emb_scale, "ce.f64": ce}
    for name, a in arrays.items():
        np.ascontiguousarray(a, dtype=np.dtype(FILES[name]).newbyteorder("<")).tofile(os.path.join(directory, "ref_" + name))
    with open(os.path.join(directory, "ref_meta.txt"), "w") as f:
        for k, v in meta.items():
            f.write("%s %s\n" % (k, v))
