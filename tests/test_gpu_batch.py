"""gfy_encode_coo_batch: several shards in one sequence of launches (include/gfy.h) must give,
shard by shard, the bytes gfy_encode_coo gives on that shard alone — the per-node arithmetic does
not depend on what else is in the launch, whichever layer kernel the launch's size selects
(one-round kernel with the fused head / persistent rounds + stand-alone head).

Reference semantics: ``encode_graphs`` runs micro-batch after micro-batch through
``_run_graph_shard`` (src/ginfinity/api.py:211-260); the reference's CPU output is invariant to
the micro-batch layout, which is what makes batching legitimate.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from ginfinity_amd import _native as native
from ginfinity_amd import synthetic

pytestmark = pytest.mark.gpu


def _device(engine, shard):
    dev = engine.device
    rows, kept = None, None
    if shard.node_roles.any():
        core = shard.node_roles == 0
        kept = int(core.sum())
        table = np.cumsum(core, dtype=np.int32) - np.int32(1)
        table[~core] = -1
        rows = torch.from_numpy(table).to(dev)
    return (torch.from_numpy(np.ascontiguousarray(shard.node_features)).to(dev),
            torch.from_numpy(np.ascontiguousarray(shard.edge_index)).to(dev),
            torch.from_numpy(np.ascontiguousarray(shard.edge_types)).to(dev), rows, kept)


@pytest.fixture(scope="module")
def mixed_shards(rouskin_shard):
    """Unequal sizes on purpose: a few hundred nodes, ~50 k nodes, a 1-node graph, a shard with
    context nodes and hubs (in-degree > 8: the direct path), the 60k/300k synthetic shard."""
    from ginfinity_amd import GraphBuilder, RNA
    tiny = GraphBuilder().build_shard([RNA("A", "A", ".")])
    return [rouskin_shard.slice(0, 4), rouskin_shard.slice(64, 400), tiny,
            synthetic.arbitrary_shard(0), synthetic.roofline_shard(3)]


def _single(engine, shard, out_dtype=torch.float16):
    x, ei, et, rows, kept = _device(engine, shard)
    return engine.encode_coo(x, ei, et, out_rows=rows, n_out=kept, out_dtype=out_dtype).cpu().numpy()


@pytest.mark.parametrize("kernel", [-1, 1, 3, 4, 5])
def test_batch_is_bit_identical_to_single_shards(gpu_encoder, mixed_shards, kernel):
    engine = gpu_encoder._engine
    try:
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, -1)
        want = [_single(engine, shard) for shard in mixed_shards]
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, kernel)
        got = engine.encode_coo_batch([_device(engine, shard) for shard in mixed_shards])
        torch.cuda.synchronize()
        for shard, a, b in zip(mixed_shards, got, want):
            assert a.shape == b.shape
            assert a.cpu().numpy().tobytes() == b.tobytes(), shard.node_count
    finally:
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, -1)


def test_batch_of_synthetic_shards_matches_the_reference_rows(gpu_encoder, golden):
    """Twelve 60k/300k shards in one call (the benchmark's batch): shards 0 and 1 against the
    rows the reference itself produced (tests/golden/synthetic.npz), the rest against the
    single-shard path."""
    engine = gpu_encoder._engine
    shards = [synthetic.roofline_shard(seed) for seed in range(12)]
    outs = engine.encode_coo_batch([_device(engine, shard) for shard in shards])
    torch.cuda.synchronize()
    g = golden("synthetic.npz")
    for seed in (0, 1):
        rows = g[f"seed{seed}.rows"]
        got = outs[seed].cpu().numpy()[rows].astype(np.float64)
        assert np.abs(got - g[f"seed{seed}.out.m16"].astype(np.float64)).max() <= 1e-3
    for seed in (2, 7, 11):
        assert outs[seed].cpu().numpy().tobytes() == _single(engine, shards[seed]).tobytes()


@pytest.mark.parametrize("kernel", [3, 4, 5])
def test_stand_alone_head_behind_persistent_rounds_gives_the_same_bytes(gpu_encoder, mixed_shards,
                                                                         kernel):
    """GFY_OPT_SEPARATE_HEAD: head + normalise as their own launch (k_head_d) behind the
    persistent-rounds layers instead of inside the last of them — the same pipeline on the
    same registers' worth of data, so the same bytes."""
    engine = gpu_encoder._engine
    inputs = [_device(engine, shard) for shard in mixed_shards]
    try:
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, kernel)
        fused = [o.cpu().numpy() for o in engine.encode_coo_batch(inputs)]
        engine.set_option(native.GFY_OPT_SEPARATE_HEAD, 1)
        apart = [o.cpu().numpy() for o in engine.encode_coo_batch(inputs)]
    finally:
        engine.set_option(native.GFY_OPT_SEPARATE_HEAD, 0)
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, -1)
    for a, b in zip(fused, apart):
        assert a.tobytes() == b.tobytes()


def test_random_batches_give_the_same_bytes_on_every_layer_kernel(gpu_encoder):
    """Twelve random batches (2-6 shards each: RNA-like records of 64 / 500 / 1,000 / 4,000
    nodes, interchange shards with hubs, context nodes and all edge types — staged tiles,
    direct-path tiles, ragged last rounds): the one-round kernel, the persistent rounds, the
    two windowed workgroups per CU and the three workgroups per CU of gine_layer_x.inc must agree
    bit for bit, shard by shard."""
    engine = gpu_encoder._engine
    rng = np.random.default_rng(4)
    try:
        for _ in range(12):
            shards = []
            for _ in range(int(rng.integers(2, 7))):
                if rng.random() < 0.3:
                    shards.append(synthetic.arbitrary_shard(
                        int(rng.integers(0, 1000)), nodes=int(rng.integers(2000, 30000)),
                        edges=int(rng.integers(5000, 120000)), records=4,
                        hub_degree=int(rng.integers(1, 60))))
                else:
                    shards.append(synthetic.roofline_shard(
                        int(rng.integers(0, 1000)), records=int(rng.integers(1, 16)),
                        length=int(rng.choice([64, 500, 1000, 4000]))))
            inputs = [_device(engine, shard) for shard in shards]
            outs = {}
            for kernel in (1, 3, 4, 5):
                engine.set_option(native.GFY_OPT_LAYER_KERNEL, kernel)
                outs[kernel] = [o.cpu().numpy() for o in engine.encode_coo_batch(inputs)]
            for kernel in (3, 4, 5):
                for a, b in zip(outs[1], outs[kernel]):
                    assert a.tobytes() == b.tobytes()
    finally:
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_batch_other_output_dtypes_equal_single(gpu_encoder, mixed_shards, dtype):
    engine = gpu_encoder._engine
    picked = mixed_shards[:4]
    got = engine.encode_coo_batch([_device(engine, s) for s in picked], out_dtype=dtype)
    torch.cuda.synchronize()
    for shard, a in zip(picked, got):
        assert a.cpu().numpy().tobytes() == _single(engine, shard, dtype).tobytes()


def test_batch_with_the_fp32_model_equals_single(gpu_encoder_fp32, mixed_shards):
    engine = gpu_encoder_fp32._engine
    picked = mixed_shards[:3]
    got = engine.encode_coo_batch([_device(engine, s) for s in picked], out_dtype=torch.float32)
    torch.cuda.synchronize()
    for shard, a in zip(picked, got):
        assert a.cpu().numpy().tobytes() == _single(engine, shard, torch.float32).tobytes()


def test_prepared_batch_step_leaves_its_workspace_ready(gpu_encoder, mixed_shards):
    """The counters of the CSR build are zero again after a call: the same pre-bound step gives
    the same bytes call after call, on a stream of the caller's choice."""
    engine = gpu_encoder._engine
    inputs = [_device(engine, shard) for shard in mixed_shards[:4]]
    outs = [torch.empty((x.shape[0] if kept is None else kept, 128), dtype=torch.float16,
                        device=engine.device) for x, _ei, _et, _rows, kept in inputs]
    step = engine.prepare_batch_step([(x, ei, et, rows, out)
                                      for (x, ei, et, rows, _k), out in zip(inputs, outs)])
    stream = torch.cuda.Stream(device=engine.device)
    step(stream.cuda_stream)
    stream.synchronize()
    first = [o.cpu().numpy().copy() for o in outs]
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    for _ in range(3):
        step(stream.cuda_stream)
    stream.synchronize()
    for a, b in zip(first, outs):
        assert a.tobytes() == b.cpu().numpy().tobytes()


def test_batch_argument_errors(gpu_encoder, mixed_shards):
    engine = gpu_encoder._engine
    one = _device(engine, mixed_shards[0])
    with pytest.raises(ValueError, match="1..16 shards"):
        engine.encode_coo_batch([])
    with pytest.raises(ValueError, match="1..16 shards"):
        engine.encode_coo_batch([one] * 17)
    lib = native.library()
    array = (native.GfyShard * 1)()
    assert lib.gfy_encode_coo_batch(engine._handle, array, 1, native.GFY_F16, 1, None, 0,
                                    None) == native.GFY_ERR_INVALID


def test_encode_graphs_issues_its_micro_batches_in_groups(gpu_encoder, golden, rouskin_shard,
                                                          monkeypatch):
    """The product path is the measured path: ``encode_graphs`` on the 897,588-node shard (15
    micro-batches, reference loop api.py:211-230) hands groups of four micro-batches to
    ``gfy_encode_coo_batch`` — the launches then run the rounds kernel, not the one-round
    kernel of a lone 60,000-node micro-batch — and gives the bytes the micro-batch-by-micro-batch
    path gives, within 1e-3 of the reference's own rows."""
    from ginfinity_amd import api
    engine = gpu_encoder._engine
    assert api.MICROBATCH_GROUP == 4
    grouped = np.concatenate(gpu_encoder.encode_graphs(rouskin_shard))
    assert engine.last_layer_kernel() == 4          # windowed rounds (gine_layer_w.inc)
    g = golden("rouskin_full.npz")
    sampled = grouped[::int(g["stride"])].astype(np.float64)
    assert np.abs(sampled - g["rows"].astype(np.float64)).max() <= 1e-3
    monkeypatch.setattr(api, "MICROBATCH_GROUP", 1)
    single = np.concatenate(gpu_encoder.encode_graphs(rouskin_shard))
    assert engine.last_layer_kernel() == 1          # a 60,000-node launch: one round per CU
    assert grouped.tobytes() == single.tobytes()
    monkeypatch.setattr(api, "MICROBATCH_GROUP", 3)  # ragged groups: 3 x 5
    assert np.concatenate(gpu_encoder.encode_graphs(rouskin_shard)).tobytes() == single.tobytes()


def test_records_and_device_blocks_take_the_grouped_path_too(gpu_encoder, rouskin_records,
                                                             rouskin_shard, monkeypatch):
    """``encode_many`` from text (graphs built on the device) and ``encode_graphs_device`` /
    ``encode_shards_device`` (what ``parallel.encode_owned_shards`` calls) issue the same
    groups; every one of them equals the micro-batch-by-micro-batch result bit for bit."""
    from ginfinity_amd import api
    engine = gpu_encoder._engine
    records = rouskin_records[:1500]                 # ~230,000 nodes: 4 micro-batches
    shard = rouskin_shard.slice(0, 1500)
    grouped = np.concatenate(gpu_encoder.encode_many(records))
    assert engine.last_layer_kernel() == 4
    block, counts = gpu_encoder.encode_graphs_device(shard)
    assert engine.last_layer_kernel() == 4
    blocks, many = gpu_encoder.encode_shards_device([shard.slice(0, 700), shard.slice(700, 1500)])
    # the resident-input form of the same thing (stage once, encode many times)
    staged, staged_counts = gpu_encoder.stage_shards(shard)
    resident = gpu_encoder.encode_staged(staged)
    assert staged_counts == [shard.core_counts]
    monkeypatch.setattr(api, "MICROBATCH_GROUP", 1)
    single = np.concatenate(gpu_encoder.encode_many(records))
    assert grouped.tobytes() == single.tobytes()
    assert block.cpu().numpy().tobytes() == single.tobytes()
    assert blocks.cpu().numpy().tobytes() == single.tobytes()
    assert resident.cpu().numpy().tobytes() == single.tobytes()
    assert counts == shard.core_counts and list(map(len, many)) == [700, 800]


def test_staged_groups_in_two_lanes_give_the_one_lane_bytes(gpu_encoder, rouskin_shard, monkeypatch):
    """``encode_staged`` keeps two groups of micro-batches in flight (two encoders, two streams:
    what ``bench.py`` times).  Same bytes as one group after the other, call after call into the
    same block, with ragged last groups, with a layer-kernel switch set after the second lane's
    encoder exists, and with work of the caller's stream in front of and behind the call."""
    from ginfinity_amd import api, _native as native
    shards = [rouskin_shard, synthetic.roofline_shard(5), synthetic.arbitrary_shard(9)]
    staged, counts = gpu_encoder.stage_shards(shards)
    assert len(staged) >= 15 and [len(c) for c in counts] == [s.record_count for s in shards]
    monkeypatch.setattr(api, "STAGED_LANES", 1)
    want = gpu_encoder.encode_staged(staged).cpu().numpy()
    monkeypatch.setattr(api, "STAGED_LANES", 2)
    block = torch.full(want.shape, float("nan"), dtype=torch.float16, device=gpu_encoder._engine.device)
    for _ in range(3):
        block.fill_(float("nan"))                       # on the caller's stream, in front of the call
        got = gpu_encoder.encode_staged(staged, out=block)
        total = got.float().sum()                       # ... and behind it
        assert got.cpu().numpy().tobytes() == want.tobytes()
        assert bool(torch.isfinite(total))
    assert gpu_encoder._lanes is not None and len(gpu_encoder._lanes) == 2
    for size in (3, 2):                                  # ragged groups, more of them
        monkeypatch.setattr(api, "MICROBATCH_GROUP", size)
        assert gpu_encoder.encode_staged(staged).cpu().numpy().tobytes() == want.tobytes()
    monkeypatch.setattr(api, "MICROBATCH_GROUP", 4)
    try:
        gpu_encoder._engine.set_option(native.GFY_OPT_LAYER_KERNEL, 3)
        assert gpu_encoder.encode_staged(staged).cpu().numpy().tobytes() == want.tobytes()
        assert [e.last_layer_kernel() for e, _stream in gpu_encoder._lanes] == [3, 3]
    finally:
        gpu_encoder._engine.set_option(native.GFY_OPT_LAYER_KERNEL, -1)
    # the encoder's other paths still work behind a two-lane call (its workspace changed streams)
    assert np.concatenate(gpu_encoder.encode_graphs(shards[1])).tobytes() == \
        want[sum(map(sum, counts[:1])):sum(map(sum, counts[:2]))].tobytes()


def test_host_shards_in_two_lanes_give_the_one_lane_bytes(gpu_encoder, rouskin_shard, monkeypatch):
    """``encode_shards_device`` (host arrays in, one device block out: what a rank of ``parallel``
    runs) can keep two groups in flight as ``encode_staged`` does (``HOST_FEED_LANES``; the default
    is one: the call is bound by the uploads), each behind its own upload.  Same
    bytes as one group after the other and as the resident form, call after call into the same
    block, with ragged groups, and with work of the caller's stream on both sides of the call."""
    from ginfinity_amd import api
    shards = [rouskin_shard, synthetic.roofline_shard(5), synthetic.arbitrary_shard(9),
              synthetic.roofline_shard(6)]
    monkeypatch.setattr(api, "HOST_FEED_LANES", 1)
    want, want_counts = gpu_encoder.encode_shards_device(shards)
    want = want.cpu().numpy()
    staged, counts = gpu_encoder.stage_shards(shards)
    assert len(staged) >= 16 and counts == want_counts
    assert gpu_encoder.encode_staged(staged).cpu().numpy().tobytes() == want.tobytes()
    monkeypatch.setattr(api, "HOST_FEED_LANES", 2)
    block = torch.full(want.shape, float("nan"), dtype=torch.float16, device=gpu_encoder._engine.device)
    for _ in range(3):
        block.fill_(float("nan"))                       # on the caller's stream, in front of the call
        got, got_counts = gpu_encoder.encode_shards_device(shards, out=block)
        total = got.float().sum()                       # ... and behind it
        assert got.cpu().numpy().tobytes() == want.tobytes() and got_counts == want_counts
        assert bool(torch.isfinite(total))
    for size in (3, 2):
        monkeypatch.setattr(api, "MICROBATCH_GROUP", size)
        assert gpu_encoder.encode_shards_device(shards)[0].cpu().numpy().tobytes() == want.tobytes()
    # an error of a packer (an edge that leaves its records) leaves the lanes usable
    broken = synthetic.roofline_shard(1, records=3, length=100)
    edges = broken.edge_index.copy()
    edges[0, 5] = 250
    object.__setattr__(broken, "edge_index", edges)
    monkeypatch.setattr(api, "MICROBATCH_GROUP", 2)     # 100-node micro-batches, pairs: 2 good groups, then the bad one
    with pytest.raises(Exception, match="edge index outside"):
        gpu_encoder.encode_shards_device([synthetic.roofline_shard(2, records=4, length=100), broken],
                                         max_batch_nodes=150)
    monkeypatch.setattr(api, "MICROBATCH_GROUP", 4)
    assert gpu_encoder.encode_shards_device(shards)[0].cpu().numpy().tobytes() == want.tobytes()


def _device_with_records(engine, shard):
    """As ``_device``, with the shard's record boundaries riding on the edge_index tensor (what
    ``Ginfinity.stage_shards`` uploads): the batch call then takes the record-range set-up."""
    from ginfinity_amd.engine import attach_records
    x, ei, et, rows, kept = _device(engine, shard)
    attach_records(ei, torch.from_numpy(np.asarray(shard.node_ptr, dtype=np.int64)).to(engine.device),
                   torch.from_numpy(np.asarray(shard.edge_ptr, dtype=np.int64)).to(engine.device))
    return x, ei, et, rows, kept


@pytest.mark.parametrize("kernel", [-1, 1, 5])
def test_record_boundaries_change_nothing_but_the_launches(gpu_encoder, mixed_shards, kernel):
    """gfy_shard.node_ptr / edge_ptr (ABI 4): COO -> tile plans by record ranges, no counting
    launch, no global atomics (csrc/csr_records.inc).  Same bytes as the counting path, shard by
    shard: RNA records (staged tiles only), interchange shards with hubs and more than 40 outside
    edges per tile (direct-path tiles: CSR rows written by the range workgroups, hub rows
    completed by the second sweep), sliced records with context rows, a one-node shard, batches
    whose shards start at non-zero tile and edge bases, and random batches."""
    engine = gpu_encoder._engine
    rng = np.random.default_rng(11)
    batches = [mixed_shards]
    for _ in range(6):
        shards = []
        for _ in range(int(rng.integers(1, 6))):
            if rng.random() < 0.4:
                shards.append(synthetic.arbitrary_shard(
                    int(rng.integers(0, 1000)), nodes=int(rng.integers(300, 20000)),
                    edges=int(rng.integers(1000, 90000)), records=int(rng.integers(1, 9)),
                    hub_degree=int(rng.integers(1, 70))))
            else:
                shards.append(synthetic.roofline_shard(
                    int(rng.integers(0, 1000)), records=int(rng.integers(1, 20)),
                    length=int(rng.choice([34, 64, 254, 258, 500, 4000]))))
        batches.append(shards)
    try:
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, kernel)
        for shards in batches:
            plain = [o.cpu().numpy() for o in
                     engine.encode_coo_batch([_device(engine, s) for s in shards])]
            ranged = [o.cpu().numpy() for o in
                      engine.encode_coo_batch([_device_with_records(engine, s) for s in shards])]
            # ... and again on the same inputs: the workspace's counters were left zero
            for a, b, shard in zip(plain, ranged, shards):
                assert a.shape == b.shape
                assert a.tobytes() == b.tobytes(), (shard.node_count, shard.record_count)
    finally:
        engine.set_option(native.GFY_OPT_LAYER_KERNEL, -1)


def test_record_ranges_of_both_sizes_give_the_counting_path_bytes(gpu_encoder):
    """The range workgroups own 512 rows (and have 512 threads) in a batch of up to 90,000 rows,
    256 rows up to 200,000 and 768 above (gfy_common.h: kRecRowsLone / kRecRowsSmall /
    kRecRowsLarge): the same shards alone and in a batch of 83,500 rows (the 512-thread
    workgroups), in one of 140,000 rows (small ranges) and in one of 203,500 rows (large ranges;
    direct-path tiles and hub rows inside them, a shard whose last range is ragged, shard bases
    that are no multiple of the range) give the bytes of the counting path."""
    engine = gpu_encoder._engine
    shards = [synthetic.roofline_shard(0), synthetic.roofline_shard(1),
              synthetic.arbitrary_shard(3, nodes=20_000, edges=90_000, records=5, hub_degree=60),
              synthetic.roofline_shard(2, records=7, length=500),
              synthetic.roofline_shard(4)]
    assert sum(s.node_count for s in shards) > 200_000
    alone = [engine.encode_coo_batch([_device(engine, s)])[0].cpu().numpy() for s in shards]
    assert sum(s.node_count for s in shards[2:]) <= 90_000 < sum(s.node_count for s in shards[:3])
    for group in ([[s] for s in shards] + [shards, shards[2:], shards[:3], shards[::-1]]):
        ranged = [o.cpu().numpy() for o in
                  engine.encode_coo_batch([_device_with_records(engine, s) for s in group])]
        for got, shard in zip(ranged, group):
            want = alone[next(i for i, s in enumerate(shards) if s is shard)]
            assert got.tobytes() == want.tobytes(), (len(group), shard.node_count)


def test_record_boundaries_and_counting_calls_share_one_workspace(gpu_encoder):
    """A call with record boundaries leaves the counting scratch untouched (zero), so calls of
    both kinds may alternate on one encoder workspace (encode_coo_group)."""
    engine = gpu_encoder._engine
    shards = [synthetic.roofline_shard(21, records=3, length=700), synthetic.arbitrary_shard(5)]
    want = [o.cpu().numpy() for o in engine.encode_coo_batch([_device(engine, s) for s in shards])]
    for with_records in (True, False, True, True, False):
        outs = [torch.empty((int(np.count_nonzero(s.node_roles == 0)), 128), dtype=torch.float16,
                            device=engine.device) for s in shards]
        make = _device_with_records if with_records else _device
        engine.encode_coo_group([(*make(engine, s)[:4], out) for s, out in zip(shards, outs)])
        torch.cuda.synchronize()
        for a, b in zip(outs, want):
            assert a.cpu().numpy().tobytes() == b.tobytes()
