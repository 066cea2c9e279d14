"""RCCL on hardware inside a 1-GPU lease: world_size 1, backend "nccl" (= RCCL on ROCm).

Runs the multi-GPU code of the path exactly as rank 0 of an N-GPU job would -- init_process_group("nccl"),
dist.broadcast_arena on the REAL MiDualAR and MiDAC arenas (in 1 MiB pieces so that many collectives are issued),
the benchmark's MAX / SUM all_reduce on float64 -- with generation interleaved, and checks that what the library
computes on ITS stream after the collectives (NCCL stream -> torch stream -> library stream ordering, dualar.hip
sync_in) equals what it computed before them.  No N > 1 curve exists for this repo: the 8-GPU runs are the driver's.
Prints RCCL_WORLD1_OK on success."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.dist import broadcast_arena, shard_utterances
    from fish_speech_amd.dual_ar import MiDualAR, generate_batch
    from oracle import dac as D
    from oracle import dual_ar as O
    from oracle.search_golden import MID

    cfg = O.DualARConfig(**MID)
    state = O.make_peaky_state(cfg, seed=93, emb_gain=2.5, slow_gain=3.0, fast_gain=2.5)
    model = MiDualAR.from_state_dict(cfg, state, device=dev, im_end_id=cfg.im_end_id)
    model.setup_caches(4, cfg.max_seq_len)
    prompts = [O.make_prompt(cfg, T, seed=s, n_semantic=ns) for T, s, ns in ((40, 3, 12), (23, 4, 0), (57, 5, 20))]
    kw = dict(max_new_tokens=24, temperature=0.7, top_p=0.7, top_k=30, seeds=[11, 12, 13], stop_on_im_end=False)
    before = generate_batch(model=model, prompts=prompts, **kw)
    dcfg = D.small_config()
    codec = MiDAC.from_state_dict(DacConfig.from_any(dcfg), D.make_synthetic_state(dcfg, seed=11), device=dev)
    codes = D.make_codes(dcfg, 2, 6, seed=2).to(dev)
    wav_before = codec.from_indices(codes.clone())

    arena_sum = int(model.arena.to(torch.int64).sum())
    for rnd in range(3):
        broadcast_arena(model, src=0, chunk_bytes=1 << 20)       # dozens of RCCL broadcasts on the real arena
        broadcast_arena(codec, src=0, chunk_bytes=1 << 20)
        t = torch.tensor([1.5 + rnd, 2.5, 100.0], device=dev, dtype=torch.float64)   # bench.py's reductions
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert mx.tolist() == t.tolist() == [1.5 + rnd, 2.5, 100.0]
        dist.barrier()
        # no host sync here on purpose: the library's stream must order itself after torch's (and NCCL's) work
        after = generate_batch(model=model, prompts=prompts, **kw)
        for a, b in zip(before, after):
            assert torch.equal(a, b), "generation changed after the RCCL collectives"
        assert torch.equal(codec.from_indices(codes.clone()), wav_before)
    assert int(model.arena.to(torch.int64).sum()) == arena_sum
    assert shard_utterances(list(range(5)), dist.get_rank(), dist.get_world_size()) == [0, 1, 2, 3, 4]
    ver = torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK backend=nccl version", ver, flush=True)


if __name__ == "__main__":
    main()
