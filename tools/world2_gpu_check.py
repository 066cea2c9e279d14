"""World size 2 on ONE GPU: the N > 1 path of SURVEY.md 8e with a real second rank.

RCCL refuses two ranks on one device, gloo does not (it moves device tensors through host memory), so inside the 1-GPU
lease two PROCESSES on cuda:0 run the multi-GPU code exactly as ranks 0 and 1 of an N-GPU job would, only the transport
differs from the xGMI run: init_process_group, rank 0 loads the S2-width Dual-AR weights and the full-size codec, rank 1
creates its handles over EMPTY arenas, dist.broadcast_arena replicates both arenas (rank 1: weights_ready, never a
load_tensor / finalize), the eight reference-written S2 utterances (tests/golden/dualar_s2_{plain,clone,ragged}.npz) are
sharded r::2 (dist.shard_utterances = tools/vqgan/extract_vq.py:207's files[RANK::WORLD_SIZE]), generated per rank with
NO data-path collective, gathered (dist.gather_results) and compared with the fixtures on rank 0; each rank decodes its
utterances' codes with its codec, rank 0 re-decodes rank 1's and requires bit equality; bench.py's MAX / SUM reductions
run across the two.  Prints WORLD2_GPU_OK (rank 0) on success.  Started without RANK in the environment it spawns the two ranks itself."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def spawn():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   GLOO_SOCKET_IFNAME=os.environ.get("GLOO_SOCKET_IFNAME", "lo"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT))
    deadline = time.time() + 330
    while time.time() < deadline and any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):   # one rank failed: the other would wait in a collective
            break
        time.sleep(0.5)
    for p in procs:
        if p.poll() is None:
            p.kill()
    rcs = [p.wait() for p in procs]
    raise SystemExit(0 if all(rc == 0 for rc in rcs) else 1)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (gloo sees HOST tensors only: its device-tensor path between ranks that share one GPU is what the GPU memory access
    # faults of profiles/r06_startup_order_stress.txt need; dist.broadcast_buffer stages through host itself)
    probe = torch.full((4,), float(rank + 1))
    dist.broadcast(probe, src=0)
    assert probe.tolist() == [1.0] * 4

    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.dist import broadcast_arena, gather_results, shard_utterances
    from fish_speech_amd.dual_ar import MiDualAR, generate_batch
    from oracle import dac as D
    from oracle import dual_ar as O

    gold = os.path.join(ROOT, "tests", "golden")

    def load(name):
        z = np.load(os.path.join(gold, f"dualar_{name}.npz"))
        skw = json.loads(str(z["state_kwargs"]))
        if "hot" in skw:
            skw["hot"] = tuple(skw["hot"])
        return z, skw

    zp, skw = load("s2_plain")
    zc, _ = load("s2_clone")
    zr, skw_r = load("s2_ragged")
    assert skw_r == skw
    ocfg = O.s2_pro_shaped_config(max_seq_len=512)
    t0 = time.time()
    model = MiDualAR(ocfg, device=dev, im_end_id=ocfg.im_end_id)
    ccfg = D.DacConfig()
    codec = MiDAC(DacConfig.from_any(ccfg), device=dev)
    if rank == 0:   # the checkpoint is read by ONE rank
        model.load_state_dict(O.make_peaky_state_hash(ocfg, device=dev, **skw))
        codec.load_state_dict(D.make_synthetic_state(ccfg, seed=3))
    else:
        assert int(model.arena[:: 1 << 20].to(torch.int64).abs().sum()) >= 0   # (touch: the arena exists, content arbitrary)
    torch.cuda.synchronize()
    t1 = time.time()
    broadcast_arena(model, src=0, chunk_bytes=1 << 28)
    broadcast_arena(codec, src=0, chunk_bytes=1 << 28)
    t2 = time.time()   # (no device synchronize: broadcast_arena orders both handles' streams after the collectives)
    if rank == 1:
        assert model.derived_info() == {"row_copies": 0, "table_rows": 0, "loaded_tensors": 0}
    model.setup_caches(4, 512)

    # the global batch of 8: every row written by the unmodified reference
    prompts, seeds, want = [None] * 8, [None] * 8, [None] * 8
    prompts[2], seeds[2], want[2] = torch.from_numpy(zp["prompt"]), int(zp["uniform_seed"]), zp["tokens"]
    prompts[5], seeds[5], want[5] = torch.from_numpy(zc["prompt"]), int(zc["uniform_seed"]), zc["tokens"]
    for row in zr["rows"].tolist():
        prompts[row], seeds[row] = torch.from_numpy(zr[f"prompt_{row}"]), int(zr["uniform_seed_base"]) + row
        want[row] = zr[f"tokens_{row}"]
    mine = shard_utterances(list(range(8)), rank, world)
    assert mine == list(range(rank, 8, world))
    out = generate_batch(model=model, prompts=[prompts[i] for i in mine], max_new_tokens=64, temperature=0.7, top_p=0.7,
                         top_k=1, seeds=[seeds[i] for i in mine], stop_on_im_end=False)
    t3 = time.time()
    info = model.derived_info()
    assert info["row_copies"] == 3 * (ocfg.n_layer + ocfg.n_fast_layer) and info["table_rows"] == ocfg.codebook_size, info
    if rank == 1:
        assert info["loaded_tensors"] == 0
    # codec on every rank: its own utterances' generated codes (the last 64 columns of rows 1..), fp32-class arithmetic
    wavs = []
    for o in out:
        codes = o[1:, -64:].unsqueeze(0).to(dev).to(torch.int64).contiguous()
        wavs.append(codec.from_indices(codes)[0])
    all_tok = gather_results(out, world, rank)
    all_wav = gather_results(wavs, world, rank)
    # bench.py's reductions (max over ranks of the elapsed time, sum of the audio seconds)
    t = torch.tensor([float(rank + 1), 2.0], dtype=torch.float64)
    mx = t.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert mx.tolist() == [2.0, 2.0] and t.tolist() == [3.0, 4.0]
    if rank == 0:
        for i in range(8):
            assert np.array_equal(all_tok[i].numpy(), want[i]), f"utterance {i} (rank {i % world}) differs from the reference"
        for i in range(8):
            if i % world == 0:
                continue
            codes = all_tok[i][1:, -64:].unsqueeze(0).to(dev).to(torch.int64).contiguous()
            again = codec.from_indices(codes)[0].cpu()
            assert torch.equal(again, all_wav[i]), f"rank {i % world}'s codec output for utterance {i} differs from rank 0's"
            assert bool(torch.isfinite(again).all()) and float(again.abs().max()) > 0
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"WORLD2_GPU_OK backend=gloo ranks=2 on cuda:0: 8/8 token matrices == reference, 4/4 rank-1 waveforms bit-equal; "
              f"load {t1 - t0:.1f} s, broadcast of both arenas {t2 - t1:.1f} s, generate {t3 - t2:.1f} s", flush=True)


if __name__ == "__main__":
    if "RANK" in os.environ:
        main()
    else:
        spawn()
