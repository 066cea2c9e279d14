"""The rank > 0 code path of SURVEY.md 8e, executed on the GPU inside the 1-GPU lease.

On ranks 1..N-1 of `configs[3]` a handle is created over an EMPTY arena, never sees load_tensor / finalize, receives the
arena bytes from `dist.broadcast_arena` and is marked ready (`weights_ready`); the fast layer-0 q|k|v table and the
row-balanced decode copies (4.1 GB at the S2 shape) are then rebuilt from the received bytes by the first prefill.
Three ways of driving that path here, each asserted against the reference-written S2 fixtures / bit-equality with the
loading handle:
  * two handles in one process: B's arena filled by a device-to-device copy of A's (the broadcast's effect);
  * bench.py's own construction order for rank 1 (`bench.construct`), with `dist.broadcast_arena`'s non-source branch
    run for real on a patched transport (the chunking, the rank test and weights_ready are the product's);
  * two PROCESSES on cuda:0 under torch.distributed (gloo transports device tensors), tools/world2_gpu_check.py.
The scheme matches the reference's tools/vqgan/extract_vq.py:161-207 (files[RANK::WORLD_SIZE], no data-path collective)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import dac as D
from oracle import dual_ar as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _replica_of(a, make):
    """What a receiving rank does: fresh handle over an empty arena <- the source's arena bytes, then weights_ready."""
    b = make()
    assert b.arena.numel() == a.arena.numel() and b.arena.data_ptr() != a.arena.data_ptr()
    b.arena.copy_(a.arena)
    b.weights_ready()   # no host synchronize: the library orders the handle's stream after the copy
    return b


def _gen(model, z, **kw):
    from fish_speech_amd.dual_ar import generate

    return generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]),
                    temperature=float(z["temperature"]), top_p=float(z["top_p"]), top_k=int(z["top_k"]),
                    seed=int(z["uniform_seed"]), **kw).numpy()


def test_dualar_replica_fed_by_the_arena_bytes_alone_generates_the_reference_tokens():
    """S2 width, bf16: B never saw a tensor; its table and its 120 row-balanced copies are rebuilt on first use; the
    full (11, 264) token matrices of B, of A and of the unmodified reference (dualar_s2_plain.npz) are equal -- alone
    and as rows of the ragged batch of 8 in which every row is the reference's."""
    from fish_speech_amd.dual_ar import MiDualAR, generate_batch
    from tests import test_s2_parity_gpu as S2

    z, skw = S2._load("s2_plain")
    cfg, A, _ = S2._model(skw)
    B = _replica_of(A, lambda: MiDualAR(cfg, device=DEV, im_end_id=cfg.im_end_id))
    B.setup_caches(8, 512)
    assert B.derived_info() == {"row_copies": 0, "table_rows": 0, "loaded_tensors": 0}
    want = z["tokens"]
    got_b = _gen(B, z)
    assert np.array_equal(got_b, want), "replica differs from the reference fixture"
    assert np.array_equal(_gen(A, z), want)
    ia, ib = A.derived_info(), B.derived_info()
    # the balanced path is the one B took: wqkv / wo / w2 of 36 + 4 layers, and all 4096 table rows
    assert ib["loaded_tensors"] == 0 and ia["loaded_tensors"] > 0
    assert ib["row_copies"] == ia["row_copies"] == 3 * (cfg.n_layer + cfg.n_fast_layer), (ia, ib)
    assert ib["table_rows"] == ia["table_rows"] == cfg.codebook_size
    B.set_graph(False)
    assert np.array_equal(_gen(B, z, poll_every=1), want), "replica, eager path"
    B.set_graph(True)
    # the ragged batch of 8 on the replica (rows 2 / 5 = s2_plain / s2_clone, the other six from dualar_s2_ragged)
    zc, _ = S2._load("s2_clone")
    zr, _ = S2._load("s2_ragged")
    prompts, seeds, wants = [None] * 8, [None] * 8, [None] * 8
    prompts[2], seeds[2], wants[2] = torch.from_numpy(z["prompt"]), int(z["uniform_seed"]), z["tokens"]
    prompts[5], seeds[5], wants[5] = torch.from_numpy(zc["prompt"]), int(zc["uniform_seed"]), zc["tokens"]
    for row in zr["rows"].tolist():
        prompts[row], seeds[row] = torch.from_numpy(zr[f"prompt_{row}"]), int(zr["uniform_seed_base"]) + row
        wants[row] = zr[f"tokens_{row}"]
    out = generate_batch(model=B, prompts=prompts, max_new_tokens=64, temperature=0.7, top_p=0.7, top_k=1, seeds=seeds,
                         stop_on_im_end=False)
    for row in range(8):
        assert np.array_equal(out[row].numpy(), wants[row]), f"replica, ragged batch row {row}"
    del B


def test_replica_filled_on_a_side_stream_with_no_host_sync_before_its_first_prefill():
    """The start-up ordering is the library's (fmi_dualar_weights_ready(h, stream)), not the harness's: the arena copy
    is enqueued on a SIDE stream behind ~50 ms of busy work, so at the time `generate` is called the bytes are certainly
    not there yet and torch's current stream knows nothing about the copy.  No synchronize anywhere between the copy
    and the first prefill (which rebuilds 4.1 GB of row-balanced copies and the q|k|v table from the arena on the
    handle's private stream).  The replica still returns the matrix the unmodified reference wrote."""
    from fish_speech_amd.dual_ar import MiDualAR
    from tests import test_s2_parity_gpu as S2

    z, skw = S2._load("s2_plain")
    cfg, A, _ = S2._model(skw)
    B = MiDualAR(cfg, device=DEV, im_end_id=cfg.im_end_id)
    B.setup_caches(2, 512)
    B.arena.zero_()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(side):
        torch.cuda._sleep(100_000_000)      # ~50 ms at 2 GHz: the copy below cannot have started when generate() runs
        B.arena.copy_(A.arena, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    B.weights_ready(stream=side)
    assert not done.query(), "the copy finished before the first call: the test would not see a missing order"
    got = _gen(B, z)
    assert np.array_equal(got, z["tokens"]), "replica read its arena before the side-stream copy landed"
    assert B.derived_info()["loaded_tensors"] == 0
    del B


def test_dualar_int8_replica_fed_by_the_arena_bytes_alone():
    """The int8 arena (int8 tiles + scales + the dequantised bf16 tiles, all inside the one blob): the replica returns
    the matrix the reference's own int8 path wrote (dualar_s2_int8.npz)."""
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR
    from tests import test_s2_parity_gpu as S2

    z, skw = S2._load("s2_int8")
    ocfg = O.s2_pro_shaped_config(max_seq_len=512)
    state = O.make_peaky_state_hash(ocfg, device=DEV, **skw)
    q = O.quantize_state_int8(ocfg, state)
    del state
    mcfg = DualARConfig.from_any(ocfg)
    mcfg.weight_int8 = True
    A = MiDualAR(mcfg, device=DEV, im_end_id=ocfg.im_end_id).load_state_dict(q)
    del q
    B = _replica_of(A, lambda: MiDualAR(mcfg, device=DEV, im_end_id=ocfg.im_end_id))
    del A
    torch.cuda.empty_cache()
    B.setup_caches(2, 512)
    got = _gen(B, z)
    assert np.array_equal(got, z["tokens"]), "int8 replica differs from the reference's int8 run"
    assert B.derived_info()["loaded_tensors"] == 0


def test_codec_replica_fed_by_the_arena_bytes_alone_is_bit_identical_in_every_arithmetic():
    """MiDAC at full size: the 16-bit operand planes and the LUTs travel inside the arena, so a replica decodes and
    encodes bit for bit like the loading handle -- fp16 split (2), fp32 matrix cores (0), bf16 / autocast (1),
    incremental decode included."""
    from fish_speech_amd.dac import DacConfig, MiDAC

    cfg = D.DacConfig()
    state = D.make_synthetic_state(cfg, seed=3)
    A = MiDAC.from_state_dict(DacConfig.from_any(cfg), state, device=DEV)
    B = _replica_of(A, lambda: MiDAC(DacConfig.from_any(cfg), device=DEV))
    codes = D.make_codes(cfg, 2, 40, seed=5).to(DEV)
    for planes in (2, 0, 1):
        A.set_precision(planes)
        B.set_precision(planes)
        wa, wb = A.from_indices(codes.clone()), B.from_indices(codes.clone())
        assert torch.equal(wa, wb), f"planes={planes}"
        assert bool(torch.isfinite(wb).all()) and float(wb.abs().max()) > 0
    A.set_precision(2)
    B.set_precision(2)
    sid = MiDAC.new_stream_id()
    parts = [B.from_indices_tail(codes[:, :, :16].clone(), 0, stream_id=sid), B.from_indices_tail(codes.clone(), 16, stream_id=sid)]
    B.close_stream(sid)
    assert torch.equal(torch.cat(parts, dim=-1), A.from_indices(codes.clone()))
    audio = torch.from_numpy(np.random.default_rng(1).standard_normal((1, 1, 3 * 44100)).astype(np.float32) * 0.1).to(DEV)
    ca, la = A.encode(audio)
    cb, lb = B.encode(audio)
    assert torch.equal(ca, cb) and torch.equal(la, lb)


def test_bench_rank1_construction_order_through_broadcast_arena(monkeypatch):
    """bench.py's own objects for rank 1 (`bench.construct`: handle, NO load, replicate, setup_caches; codec likewise)
    with dist.broadcast_arena's non-source branch executed for real: only the transport is patched (the broadcast of a
    piece copies it from rank 0's arena).  One benchmark step on the replica equals rank 0's step: codes and waveform."""
    import torch.distributed as dist

    import bench
    from fish_speech_amd.dist import broadcast_arena

    cfg = bench.s2_pro_config()
    dev = torch.device(DEV)
    m0, c0, _, _ = bench.construct(cfg, dev, 0)
    sources, calls = {}, []

    def fake_broadcast(t, src=0, group=None):
        arena = sources["cur"]
        off = t.data_ptr() - sources["dst"].data_ptr()
        assert 0 <= off and off + t.numel() <= arena.numel()
        t.copy_(arena[off: off + t.numel()])
        calls.append(t.numel())

    monkeypatch.setattr(dist, "broadcast", fake_broadcast)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: 1)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")   # the device-tensor path, as over RCCL
    order = iter([m0, c0])

    def replicate(obj):
        sources["cur"], sources["dst"] = next(order).arena, obj.arena
        broadcast_arena(obj, src=0, chunk_bytes=1 << 28)

    m1, c1, s1, cs1 = bench.construct(cfg, dev, 1, replicate=replicate)
    assert s1 is None and cs1 is None and len(calls) > 30          # 9.15 GB + the codec arena in 256 MiB pieces
    assert m1.derived_info()["loaded_tensors"] == 0
    prompts = bench.make_prompts(cfg, 4, 1000)
    seeds = [4242 + i for i in range(4)]
    frames = bench.N_FRAMES
    try:
        bench.N_FRAMES = 40
        codes0, wav0 = bench.run_step(m0, c0, prompts, seeds, dev)
        codes1, wav1 = bench.run_step(m1, c1, prompts, seeds, dev)
    finally:
        bench.N_FRAMES = frames
    assert torch.equal(codes0, codes1) and torch.equal(wav0, wav1)
    assert m1.derived_info()["row_copies"] == m0.derived_info()["row_copies"] > 0
    assert m1.derived_info()["table_rows"] == cfg.codebook_size


def test_two_processes_on_one_gpu_under_torch_distributed():
    """World size 2 on cuda:0 (gloo; dist.broadcast_buffer stages device tensors through host for it): rank 0 loads, rank 1 receives both arenas through
    dist.broadcast_arena and has never seen a tensor; the eight reference-written S2 utterances are sharded r::2
    (dist.shard_utterances), generated per rank, gathered (dist.gather_results) and compared with the fixtures; the
    codec decodes on both ranks agree bit for bit; bench.py's MAX / SUM reductions run across the two."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "world2_gpu_check.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode(errors="replace")
    print(out[-3000:])
    assert r.returncode == 0 and "WORLD2_GPU_OK" in out, out[-6000:]


def test_bench_py_runs_its_own_two_rank_path_on_one_gpu():
    """bench.py --gpus 2 end to end in its one-GPU debug mode (FMI_BENCH_ONE_GPU=1: both ranks on cuda:0, gloo instead of
    RCCL): the re-exec under torch.distributed.run, rank 1's construction through dist.broadcast_arena (no tensor ever
    loaded there), the barriers around the timed region, the MAX reduction of the elapsed time and rank 0's ONE JSON
    line with n_gpus = 2 and twice the audio of a one-rank step.  The ranks time-slice the GPU, so the line marks itself
    invalid as a result; what is checked is that the N > 1 code of the benchmark runs and accounts correctly."""
    import json

    env = dict(os.environ, FMI_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode(errors="replace")
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, out[-4000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["data"].startswith("INVALID AS A RESULT")
    assert d["config"]["parallelism"] == "utterance-sharded x2" and d["config"]["batch_per_gpu"] == 8
    audio = 2 * 8 * 215 * 2048 / 44100                       # both ranks' utterances count
    assert abs(d["value"] - audio / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["breakdown_ms"]["launches_per_frame"] == 374 and 0 < d["roofline"]["frac"] < 1
