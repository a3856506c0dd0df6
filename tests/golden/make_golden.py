"""Generate the committed golden fixtures from the GENUINE reference.

Run only in the build container, where the reference is mounted read-only:

    python tests/golden/make_golden.py            # writes tests/golden/*.npz|json

It imports ``ginfinity`` from /root/reference/src (never copied into this
repo, never shipped to the GPU box) and records, for seeded / bundled inputs,
what the reference's own CPU encode returns.  The fixtures are data only:
inputs, expected outputs, per-stage activations captured with forward hooks.

Fixture list:
  example8.npz      F1  8-nt README example: inputs, outputs for fp16/fp32 model
                        and f16/f32/f64 embedding dtype, per-stage tensors
  sliced.npz        F2  GGGAAACCCUUUUGGG window [9,16), hops 1/2/3: graph arrays
                        + outputs (core rows only)
  degenerate.npz    F3  A/. , AC/.. , GC/() graphs + outputs
  rouskin64.npz     F4a first 64 records of rouskin_sample_6k.tsv: full fp16
                        output; fp32-model output of the first 16; per-stage
                        tensors of the first 4
  rouskin_full.npz  F4b whole shard (897,588 nodes): every 97th output row
  synthetic.npz     F5  roofline_shard(seed 0/1): 1,024 sampled output rows each
  arbitrary.npz     F6  arbitrary_shard(seed 0): every 4th output row
  integers.json     F7  SHA-256 of the reference builder's arrays for the whole
                        rouskin shard, micro-batch boundaries at default limits
"""
from __future__ import annotations

import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, str(ROOT))

import ginfinity as ref                      # noqa: E402  (the genuine reference)
from ginfinity_amd import synthetic           # noqa: E402  (seeded generators)

TSV = HERE / "rouskin_sample_6k.tsv"
ARRAYS = ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr",
          "residue_index", "node_roles")


def to_reference_shard(shard) -> "ref.GraphShard":
    return ref.GraphShard(
        identifiers=shard.identifiers, sequences=shard.sequences,
        structures=shard.structures, node_features=shard.node_features,
        edge_index=shard.edge_index, edge_types=shard.edge_types,
        node_ptr=shard.node_ptr, edge_ptr=shard.edge_ptr,
        spec=ref.GraphSpec.bundled(), residue_index=shard.residue_index,
        node_roles=shard.node_roles)


def traced_encode(encoder, shard) -> dict[str, np.ndarray]:
    """Per-stage activations of one forward, named as oracle/gine_numpy.py."""
    model = encoder._model
    got: dict[str, np.ndarray] = {}
    handles = []

    def grab(name, which="out"):
        def hook(_module, inputs, output):
            value = output if which == "out" else inputs[0]
            got[name] = value.detach().cpu().numpy().copy()
        return hook

    handles.append(model.input.register_forward_hook(grab("h0")))
    for l, conv in enumerate(model.convs):
        handles += [
            conv.mlp[0].register_forward_hook(grab(f"l{l}.z", "in")),
            conv.mlp[0].register_forward_hook(grab(f"l{l}.u")),
            conv.mlp[2].register_forward_hook(grab(f"l{l}.v")),
            conv.mlp[4].register_forward_hook(grab(f"l{l}.w")),
            model.norms[l].register_forward_hook(grab(f"l{l}.y")),
        ]
        if l + 1 < len(model.convs):
            handles.append(model.convs[l + 1].register_forward_hook(
                grab(f"l{l}.h", "in")))
    handles.append(model.head[0].register_forward_hook(
        grab(f"l{len(model.convs) - 1}.h", "in")))
    handles.append(model.head[1].register_forward_hook(grab("head.t")))
    handles.append(model.head[2].register_forward_hook(grab("o")))
    encoder.encode_graphs(shard)
    for handle in handles:
        handle.remove()
    return got


def cat(outputs) -> np.ndarray:
    return np.concatenate(outputs, axis=0)


def sha(array: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(array).tobytes()).hexdigest()


def main() -> None:
    torch.manual_seed(0)
    enc16 = ref.Ginfinity.load()
    enc32 = ref.Ginfinity.load(full_precision=True)
    builder = ref.GraphBuilder()

    # ---- F1 ---------------------------------------------------------------
    record = ref.RNA("example", "ACGUACGU", "((....))")
    shard = builder.build_shard([record])
    fixture = {name: getattr(shard, name) for name in ARRAYS}
    for tag, encoder in (("m16", enc16), ("m32", enc32)):
        for dtype in ("float16", "float32", "float64"):
            fixture[f"out.{tag}.{dtype}"] = encoder.encode(
                record, embedding_dtype=dtype)
    fixture.update({f"stage.{k}": v
                    for k, v in traced_encode(enc16, shard).items()})
    np.savez_compressed(HERE / "example8.npz", **fixture)

    # ---- F2 ---------------------------------------------------------------
    sequence, structure = "GGGAAACCCUUUUGGG", "......(((....)))"
    fixture = {}
    for hops in (1, 2, 3):
        windowed = ref.RNA("stem", sequence, structure, start=9, end=16)
        graph = ref.GraphBuilder(
            keep_paired_neighbours=True, context_hops=hops).build(windowed)
        for name in ("node_features", "edge_index", "edge_types",
                     "residue_index", "node_roles"):
            fixture[f"hops{hops}.{name}"] = getattr(graph, name)
        fixture[f"hops{hops}.out.m16"] = enc16.encode(
            windowed, keep_paired_neighbours=True, context_hops=hops)
        fixture[f"hops{hops}.out.m32.float32"] = enc32.encode(
            windowed, keep_paired_neighbours=True, context_hops=hops,
            embedding_dtype="float32")
    fixture["nokeep.out.m16"] = enc16.encode(
        ref.RNA("stem", sequence, structure, start=9, end=16))
    np.savez_compressed(HERE / "sliced.npz", **fixture)

    # ---- F3 ---------------------------------------------------------------
    fixture = {}
    for seq, struct in (("A", "."), ("AC", ".."), ("GC", "()")):
        rec = ref.RNA(seq, seq, struct)
        graph = builder.build(rec)
        fixture[f"{seq}.edge_index"] = graph.edge_index
        fixture[f"{seq}.edge_types"] = graph.edge_types
        fixture[f"{seq}.node_features"] = graph.node_features
        fixture[f"{seq}.out.m16"] = enc16.encode(rec)
        fixture[f"{seq}.out.m32.float32"] = enc32.encode(
            rec, embedding_dtype="float32")
    np.savez_compressed(HERE / "degenerate.npz", **fixture)

    # ---- F4 ---------------------------------------------------------------
    records = ref.read_rna_table(TSV)
    full = builder.build_shard(records)
    first64 = full.slice(0, 64)
    fixture = {"out.m16": cat(enc16.encode_graphs(first64)),
               "out.m32.float32": cat(enc32.encode_graphs(
                   full.slice(0, 16), embedding_dtype="float32"))}
    fixture.update({f"stage.{k}": v for k, v in
                    traced_encode(enc16, full.slice(0, 4)).items()})
    np.savez_compressed(HERE / "rouskin64.npz", **fixture)

    whole = cat(enc16.encode_graphs(full))
    np.savez_compressed(HERE / "rouskin_full.npz",
                        stride=np.int64(97), rows=whole[::97])

    # ---- F5 ---------------------------------------------------------------
    fixture = {}
    for seed in (0, 1):
        syn = synthetic.roofline_shard(seed)
        out = cat(enc16.encode_graphs(to_reference_shard(syn)))
        pick = np.sort(np.random.default_rng(1000 + seed).choice(
            syn.node_count, size=1024, replace=False))
        fixture[f"seed{seed}.rows"] = pick
        fixture[f"seed{seed}.out.m16"] = out[pick]
        if seed == 0:
            out32 = cat(enc32.encode_graphs(
                to_reference_shard(syn), embedding_dtype="float32"))
            fixture["seed0.out.m32.float32"] = out32[pick]
    np.savez_compressed(HERE / "synthetic.npz", **fixture)

    # ---- F6 ---------------------------------------------------------------
    arb = synthetic.arbitrary_shard(0)
    arb_ref = to_reference_shard(arb)
    core = arb.node_roles == 0
    out = cat(enc16.encode_graphs(arb_ref))            # core rows only
    out32 = cat(enc32.encode_graphs(arb_ref, embedding_dtype="float32"))
    assert out.shape[0] == int(core.sum())
    np.savez_compressed(HERE / "arbitrary.npz", stride=np.int64(4),
                        **{"out.m16": out[::4], "out.m32.float32": out32[::4]})

    # ---- F7 ---------------------------------------------------------------
    limits = (60_000, 300_000)
    bounds, start = [], 0
    lengths, edge_counts = full.lengths, full.edge_counts
    while start < full.record_count:            # what api.py:211-230 produces
        stop, nodes, edges = start, 0, 0
        while stop < full.record_count:
            if stop > start and (nodes + lengths[stop] > limits[0]
                                 or edges + edge_counts[stop] > limits[1]):
                break
            nodes += lengths[stop]
            edges += edge_counts[stop]
            stop += 1
        bounds.append([start, stop, nodes, edges])
        start = stop
    # cross-check the restated loop against the reference's own slicing
    seen = []
    original = enc16._run_graph_shard
    enc16._run_graph_shard = lambda s, d: (seen.append(
        [s.record_count, s.node_count, s.edge_count]) or original(s, d))
    enc16.encode_graphs(full.slice(0, 1300))
    enc16._run_graph_shard = original
    assert seen[:3] == [[b[1] - b[0], b[2], b[3]] for b in bounds[:3]], seen[:3]
    integers = {
        "rouskin": {
            "records": full.record_count, "nodes": full.node_count,
            "edges": full.edge_count,
            "sha256": {name: sha(getattr(full, name)) for name in ARRAYS},
            "microbatches_60000_300000": bounds,
        },
        "graph_spec_sha256": full.spec.sha256,
        "versions": {"torch": torch.__version__, "numpy": np.__version__,
                     "reference": ref.__version__},
    }
    (HERE / "integers.json").write_text(json.dumps(integers, indent=1) + "\n")
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
