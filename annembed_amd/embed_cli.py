"""Command line of the reference's `embed` binary (src/bin/embed.rs:224-321 flags, :330-445 flow) on the HIP library, so
that embeddings can be produced from the same CSV files and diffed against the reference's output.

    python -m annembed_amd.embed_cli --csv data.csv [--out embedded.csv] [--delim ,] [--batch 20] [--stepg 2.]
        [--nbsample 10] [--layer 0] [--scale 1.0] [--dim 2] [--quality f]
        [hnsw --dist DistL2 --nbconn 64 --ef 512 --knbn 10 [--scale_modify_f 1.0]]

Same flags, defaults and output format.  Differences, all on the graph producer (HNSW is un-vendored third-party code,
out of scope): the kNN graph is the EXACT L2 graph of `ae_kgraph_bruteforce_l2` (`hnsw --nbconn/--ef/--scale_modify_f`
are accepted and unused; `--dist` other than DistL2 is refused), and `--layer > 0` (hierarchical init from an HNSW
layer, embed.rs:418-433) is refused because there is no layer structure to project from.  `--stepg` is accepted and
not applied -- exactly as in the reference, whose parse_embed_group (embed.rs:141-158) never reads it.  The reference
reads the CSV as f64; the library computes in f32.
"""
import argparse
import sys

import numpy as np


def build_parser():
    ap = argparse.ArgumentParser(prog="annembed", description="Non-linear Dimension Reduction/Embedding via Approximate Nearest "
                                 "Neighbor Graph, HNSW Initialization")
    ap.add_argument("--csv", dest="csvfile", required=True, help="Expecting a csv file")
    ap.add_argument("--out", "-o", dest="outfile", default=None, help="Output file name")
    ap.add_argument("--delim", dest="delim", default=None, help="Delimiter can be ' ', ','")
    ap.add_argument("--batch", type=int, default=20, help="Number of batches to run")
    ap.add_argument("--stepg", type=float, default=2.0, help="Scale of gradient steps")
    ap.add_argument("--nbsample", type=int, default=10, help="Number of edge sampling")
    ap.add_argument("--layer", "-l", dest="hierarchy", type=int, default=0, help="A layer num")
    ap.add_argument("--scale", type=float, default=1.0, help="Spatial scale factor")
    ap.add_argument("--dim", dest="dimension", type=int, default=2, help="Dimension of embedding")
    ap.add_argument("--quality", "-q", type=float, default=None, help="Sampling fraction, should <= 1.")
    sub = ap.add_subparsers(dest="subcmd")
    h = sub.add_parser("hnsw", help="Build HNSW graph")
    h.add_argument("--dist", "-d", required=True, help='one of "DistL1", "DistL2", "DistCosine", "DistJeffreys"')
    h.add_argument("--nbconn", type=int, required=True, help="Maximum number of build connections allowed (M in HNSW)")
    h.add_argument("--ef", type=int, required=True, help="Build factor ef_construct in HNSW")
    h.add_argument("--scale_modify_f", dest="scale_modification", type=float, default=1.0)
    h.add_argument("--knbn", type=int, required=True, help="Number of k-nearest neighbours to be retrieved for embedding")
    return ap


def parse(argv):
    """-> (namespace with the embed flags, dict of HnswParams) ; HnswParams::my_default (embed.rs:66-74) without `hnsw`"""
    ns = build_parser().parse_args(argv)
    hnsw = {"max_conn": 64, "ef_c": 512, "knbn": 10, "distance": "DistL2", "scale_modification": 1.0}
    if ns.subcmd == "hnsw":
        if ns.dist not in ("DistL2", "DistL1", "DistCosine", "DistJeffreys"):
            raise SystemExit("not a valid distance")  # embed.rs:134
        hnsw = {"max_conn": ns.nbconn, "ef_c": ns.ef, "knbn": ns.knbn, "distance": ns.dist,
                "scale_modification": ns.scale_modification}
    if ns.delim is not None and len(ns.delim) != 1:
        raise SystemExit("--delim expects one character")
    return ns, hnsw


def main(argv=None):
    ns, hnsw = parse(sys.argv[1:] if argv is None else argv)
    import annembed_amd as A
    from annembed_amd import io

    if hnsw["distance"] != "DistL2":
        raise SystemExit("only DistL2 graphs are produced on the device (exact kNN); got %s" % hnsw["distance"])
    if ns.hierarchy != 0:
        raise SystemExit("--layer > 0 needs an HNSW layer structure (src/bin/embed.rs:418-433); not available without hnsw_rs")
    params = A.EmbedderParams()  # EmbedderParams::default, then the five fields parse_embed_group sets (embed.rs:149-153)
    params.nb_grad_batch = ns.batch
    params.asked_dim = ns.dimension
    params.scale_rho = ns.scale
    params.nb_sampling_by_edge = ns.nbsample
    params.hierarchy_layer = ns.hierarchy
    fraction = ns.quality if ns.quality is not None else 1.0
    data = io.get_toembed_from_csv(ns.csvfile, ns.delim if ns.delim is not None else ",", fraction)  # embed.rs:385
    if data.shape[0] <= hnsw["knbn"]:
        raise SystemExit("not enough records (%d) for knbn = %d" % (data.shape[0], hnsw["knbn"]))
    out = ns.outfile if ns.outfile is not None else "embedded.csv"  # embed.rs:369-374
    kgraph = A.KGraph.bruteforce_l2(np.ascontiguousarray(data, np.float32), hnsw["knbn"])
    embedder = A.Embedder(kgraph, params)
    if embedder.embed() != 1:
        raise SystemExit("embedding failed")  # embed.rs:407-410
    io.write_csv_array2(out, embedder.get_embedded_reindexed())  # embed.rs:413
    if ns.quality is not None:
        print(embedder.get_quality_estimate_from_edge_length(100))  # embed.rs:416-418
    return 0


if __name__ == "__main__":
    sys.exit(main())
