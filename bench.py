#!/usr/bin/env python3
"""bench.py -- poses/sec of the IFFNeRF per-query hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N = 1: plain process; N > 1: launched by torch.distributed.run)

Step = one pass of the hot path over one batch of synthetic queries (`--batch`, default 16 query images, EACH with its own
freshly drawn ray set), COLD: every step re-runs for every query of the batch stage A (device-side
surface sampler + normals + 27-ray fans), stage B (20-sample VM march + Ref shading of every ray) and stage C (ray
encoder + q/k projections (folded, include/iffnerf_hip.h), softmax over rays, column-sum score, top-100, closed-form pose).  Nothing is cached between
steps except the model tables; the ray encoder is recomputed per step as the reference does per image
(pose_estimation/identification_module.py:164).  Workload at N = 1 is BASELINE.json configs[1]: "lego 800x800, 16k
candidate rays": a synthetic lego-shaped TensorVMSplit (300^3 grid, 16/48 components, 180^3 mask), gen_points = 593 ->
16 011 rays per query, M = 256 image tokens per query (the 800x800 image only feeds the out-of-path DINOv2 front end).
`value` counts poses: batch x steps / time.
At N > 1 the same ray set is sharded over the ranks (contiguous blocks of surface points) and each step processes N
queries (weak scaling: one more query per step per GPU), with the two RCCL all_gathers of iffnerf_amd/distributed.py.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel (measured live
with events on the launch stream) and `cpu_baseline` (the oracle, i.e. the reference's PyTorch-CPU op chain, timed on
this box's host cores on a bounded sample: one cold pose at the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GEN_POINTS = 593          # -> 16 011 rays (27 per surface point)
M_TOKENS = 256
TOPK = 100
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
L2_GATHER_GBS = 17800.0   # MI355X_MICROARCH.md "Indexed rows": rows served from the XCDs' L2 gather at 16.8-18.8 TB/s chip-wide
# fused encoder/logits launch: bf16 MFMA flops per ray = 6 products x 2 x 256 outputs x (144 + 144 + 256 + 256 encoder k
# (the 141 input columns in 9 k-steps of 16, twice: layer 1 and the x-part of layer 3) + 256 logits k per 256-token block)
def trunk_flops(n_rays, m_tokens):
    return n_rays * 12.0 * 256.0 * (144 + 144 + 256 + 256 + 256 * ((m_tokens + 255) // 256))
# march: algorithmic bytes per sample = valid*1184 (8 mask bytes x4 + density taps) + shaded*3456 (appearance taps)
# (SURVEY.md section 8d, fp32 tables); per-ray terms are added where the launches read / write them
B_VALID, B_APP = 32 + 1152, 3456


def build_inputs(device, grid=300):
    from iffnerf_amd import synthetic
    from iffnerf_amd.pipeline import PosePipeline
    mask = max(32, int(round(grid * 0.6)))
    ck = synthetic.make_field_ckpt(grid=(grid, grid, grid), mask_res=(mask, mask, mask), seed=1234, step_ratio=0.5, peak=20.0)
    idw = synthetic.make_id_weights(seed=99)
    pipe = PosePipeline.from_checkpoints(ck, idw, device, model_up=(0.0, 0.0, 1.0))
    return ck, idw, pipe


def cpu_baseline(ck, idw, tokens_cpu, max_seconds=40.0):
    """Reference CPU path (oracle = the reference's op chain on torch-CPU) on ONE cold pose of the same workload."""
    from oracle import emit as oemit, field as ofield, identify as oid, pose as opose
    # the box's CPU share for one GPU is 16 cores; torch with one thread per *visible* core (256 here) is an order of
    # magnitude slower on this op mix (thread fan-out on small tensors), which would flatter the GPU number
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    f = ofield.field_from_ckpt(ck)
    torch.manual_seed(55176280)
    t0 = time.perf_counter()
    n = 0
    stage = {}
    while True:
        ta = time.perf_counter()
        o, d, c = oemit.explore_model(f, gen_points=GEN_POINTS)
        tb = time.perf_counter()
        idx, val, _, _ = oid.test_image(idw, tokens_cpu, o, d, c, TOPK)
        opose.pose_from_topk(idx, val, o, d, torch.tensor([0.0, 0.0, 1.0]))
        tc = time.perf_counter()
        n += 1
        stage = {"emit_s": tb - ta, "identify_pose_s": tc - tb}
        if tc - t0 > 10.0 or n >= 3 or (tc - t0) + (tc - ta) > max_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "poses/s", "cores": threads, "kind": "port",
            "sample": f"{n} cold pose(s) of the same workload (gen_points={GEN_POINTS}, 16011 rays, M={M_TOKENS}); "
                      f"last: emission {stage['emit_s']:.2f} s, identification+pose {stage['identify_pose_s']:.3f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--in-flight", type=int, default=4, help="cold queries kept in flight on separate streams (N = 1)")
    ap.add_argument("--batch", type=int, default=16,
                    help="cold queries per step at N = 1, each with its own freshly drawn ray set, served by one set of launches")
    ap.add_argument("--grid", type=int, default=300,
                    help="side of the synthetic VM grid; 300 is the BASELINE lego-sized model (71 MB of tables, Infinity-Cache "
                         "resident), 640 a 320 MB model whose gathers go to HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefetch", action="store_true",
                    help="draw the next query's surface points in a parallel branch of each query graph (shorter single-stream "
                         "latency, 0.40 vs 0.54 ms; no gain with several graphs in flight: ROCm 7.2 serialises the branches of "
                         "concurrently launched graphs)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N > 1 code path (captured segments + all_gathers) at any world size, for rehearsal on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE={world_size})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world_size > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=device)

    from iffnerf_amd import synthetic
    ck, idw, pipe = build_inputs(device, args.grid)
    sharded = world_size > 1 or args.force_sharded
    Q = world_size if sharded else max(1, args.batch)
    tokens = torch.stack([synthetic.make_tokens(M_TOKENS, 384, seed=7 + q) for q in range(Q)]).to(device)

    def barrier():
        torch.cuda.synchronize(device)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize(device)

    # `in_flight` steps are kept in flight, each a captured hipGraph (N = 1) or three captured segments with the two RCCL
    # all_gathers issued eagerly between them (N > 1), replayed round-robin on their own streams, so the latency-bound
    # surface sampler of one step (47 workgroups) overlaps the throughput-bound stages of another.  Every replay bumps a
    # device-side counter that is added to the sampler seed: no two steps draw the same rays.  At N > 1 all steps use the
    # one default process group, so every rank issues the collectives in the same order.
    in_flight = max(1, args.in_flight)
    # the persistent samplers of all in-flight steps must be co-resident (their workgroups meet at in-kernel barriers)
    in_flight = min(in_flight, pipe.max_steps_in_flight(GEN_POINTS, 1 if (world_size > 1 or args.force_sharded) else max(1, args.batch)))
    streams = [torch.cuda.Stream(device=device) for _ in range(in_flight)]
    if not sharded and Q > 1:
        graphs = [pipe.capture_query_batch(tokens.shape, GEN_POINTS, seed=(g + 1) << 40, k=TOPK) for g in range(in_flight)]
        for g in graphs:
            g.tokens.copy_(tokens)
    elif not sharded:
        graphs = [pipe.capture_query(tokens[0].shape, GEN_POINTS, seed=(g + 1) << 40, k=TOPK,
                                     prefetch_emission=args.prefetch) for g in range(in_flight)]
        for g in graphs:
            g.tokens.copy_(tokens[0])
    else:
        try:
            graphs = [pipe.capture_query_sharded(tokens.shape, GEN_POINTS, seed=(g + 1) << 40, k=TOPK) for g in range(in_flight)]
            for g in graphs:
                g.tokens.copy_(tokens)
            launch_mode = "3 hipGraph segments + 2 eager RCCL all_gathers per step"
        except Exception as exc:      # capture refused on this stack: same step, eager launches, one stream (all ranks alike:
            graphs = None             # capture problems are deterministic properties of the software stack)
            in_flight = 1
            launch_mode = f"eager (segment capture failed: {type(exc).__name__})"
            print(f"[bench] rank {rank}: segment capture failed ({exc!r}); running the sharded step eagerly", file=sys.stderr)
    torch.cuda.synchronize(device)

    def step(i):
        if graphs is None:
            return pipe.query_sharded(tokens, GEN_POINTS, seed=1000 + i, k=TOPK, materialize_map=False)[0]
        with torch.cuda.stream(streams[i % in_flight]):
            return graphs[i % in_flight].replay()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if sharded:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out).all()

    result = None
    if rank == 0:
        # ---- per-stage and dominant-kernel timing with events on the launch stream (outside the timed region)
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
        n_rep = 20
        stage_ms = {"sampler": 0.0, "normals_emit": 0.0, "march": 0.0, "encoder_logits": 0.0, "score_topk_pose": 0.0}   # per launch set of QB queries, eager
        march_launch_ms = [0.0, 0.0, 0.0]      # K4a density+compositing, K4b appearance gather, K4c Ref shading
        trunk_ms = []                          # k5_trunk<true>: encoder + logits + softmax partials
        bytes_a = bytes_b = 0.0
        from iffnerf_amd import hip_identify as H
        from iffnerf_amd.hip_field import isocell_emit
        # the instrumented launches have the shape of the timed region's: QB queries per launch (QB = --batch at N = 1)
        QB = Q if not sharded else 1
        tokb = tokens[:QB].reshape(QB * M_TOKENS, -1).contiguous()
        for r in range(n_rep):
            e = [ev() for _ in range(6)]
            e[0].record()
            samples, _, _ = pipe.field.surface_sample_batched(QB, GEN_POINTS, pipe.rho, 4, 200, seed=5000 + r)
            e[1].record()
            samples = samples.reshape(QB * GEN_POINTS, 3)
            normals = pipe.field.point_normals(samples)
            ori, dirs, rays = isocell_emit(pipe.cells, samples, normals, want_rays6=True)
            e[2].record()
            rgb = pipe.field.march(rays, 0, 20, want_alpha=False)[0]
            e[3].record()
            qf = pipe.idnet.q_fold(tokb)
            logits, rmax, rsum = pipe.idnet.ray_logits_folded_batched(qf, ori, dirs, rgb, QB)   # fused trunk/logits + stats merge
            e[4].record()
            score = H.attn_colsum_batched(logits, rmax, rsum, QB, write_attention=False)
            idx, val = H.topk_batched(score, TOPK)
            H.pose_from_topk_batched(idx, val, ori.view(QB, -1, 3), dirs.view(QB, -1, 3), pipe.model_up)
            e[5].record()
            torch.cuda.synchronize(device)
            for name, a, b in zip(stage_ms, e[:-1], e[1:]):
                stage_ms[name] += a.elapsed_time(b) / n_rep
            # the same march again through the instrumented entry point: per-launch durations from events on the launch
            # stream, and the kernels' own (valid, shaded) sample counters for the algorithmic byte count
            ms = []
            counts = pipe.field.march(rays, 0, 20, want_alpha=False, want_counts=True, stage_ms=ms)[4].double().sum(0)
            for i in range(3):
                march_launch_ms[i] += ms[i] / n_rep
            pipe.idnet.ray_logits_folded_batched(qf, ori, dirs, rgb, QB, trunk_ms=trunk_ms)
            R = rays.shape[0]
            bytes_a += (R * (24 + 8 + 20 * 4) + counts[0].item() * B_VALID) / n_rep           # rays in, acc/depth + weights out
            bytes_b += (R * (24 + 20 * 4 + 28 * 4) + counts[1].item() * B_APP) / n_rep          # rays + weights in, features out
        # ---- roofline of the dominant kernel (largest share of GPU time in profiles/r01_bench_kernel_stats_v4*.csv):
        # k5_trunk<true>, the fused ray encoder + attention logits, bound by the bf16 matrix cores.  The two gather
        # kernels of the march follow in `other_kernels` with their HBM-side view.
        try:   # HBM-side bytes per launch from the committed PMC passes (profiles/README.md), not measured live
            with open(os.path.join(ROOT, "profiles", "r01v5_hbm_traffic.json")) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            pmc = {}
        def traffic(*keys):
            for key in keys:
                if key in pmc:
                    return pmc[key].get("hbm_bytes_per_launch")
            return None
        n_rays = QB * GEN_POINTS * 27
        t_ms = sum(trunk_ms) / max(len(trunk_ms), 1)
        issued = QB * trunk_flops(GEN_POINTS * 27, M_TOKENS)                    # bf16 MFMA flops the kernel issues
        tf = issued / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
        # the reference's fp32 chain is 898 048 FLOP per ray for the encoder + 2*384*M for the logits
        ref_flops = n_rays * (898048.0 + 2.0 * 384.0 * M_TOKENS)
        dom_gbs = bytes_b / (march_launch_ms[1] * 1e-3) / 1e9
        roofline = {
            "kernel": "k5_trunk<true> (ray encoder + attention logits + softmax partials, one launch)",
            "queries_per_launch": QB, "rays_per_launch": n_rays,
            "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic("k5_trunk<true, 1>", "k5_trunk<true, 2>", "k5_trunk<true>"),
            "avg_launch_ms": round(t_ms, 4), "flops_per_launch": round(issued),
            "note": "achieved = bf16 MFMA flops issued / launch time: 12 x 256 x (800 encoder + 256 logits k) per ray, i.e. the "
                    "folded algorithm (2 x 256 x 1056 = 541 kFLOP per ray) times the 6 bf16 products that make one "
                    "fp32-accurate product; peak = dense bf16.  The reference's unfolded fp32 chain (SURVEY 8d) would be "
                    "%.1f GFLOP per launch = %.0f TFLOP/s at this duration (fp32-MFMA peak: 157)"
                    % (ref_flops / 1e9, ref_flops / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0),
            "traffic_source": "profiles/r01v5_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2x FETCH correction)",
            "other_kernels": {
                "k4b_appearance (appearance gather of TensorBase.forward)": {
                    "bound": "hbm", "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic("k4b_appearance<27, true, 1>", "k4b_appearance<27, true>", "k4b_appearance<27>"),
                    "algorithmic_bytes_per_launch": round(bytes_b), "avg_launch_ms": round(march_launch_ms[1], 4),
                    "l2_gather_peak": L2_GATHER_GBS, "frac_of_l2_gather_peak": round(dom_gbs / L2_GATHER_GBS, 4),
                    "note": "3456 B per shaded sample (SURVEY 8d) x the kernel's own shaded-sample counter; the tables (71 MB) "
                            "are L2 / Infinity-Cache resident, so only `traffic` bytes cross the L2's memory side and the HBM "
                            "fraction exceeds 1; the roof that binds is the rate at which the CUs can gather rows out of "
                            "the XCDs' L2s (16.8-18.8 TB/s chip-wide, MI355X_MICROARCH.md 'Indexed rows')"},
                "k4a_density_composite": {
                    "bound": "hbm", "achieved": round(bytes_a / (march_launch_ms[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(bytes_a / (march_launch_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": traffic("k4a_density_composite<1>", "k4a_density_composite"), "algorithmic_bytes_per_launch": round(bytes_a),
                    "avg_launch_ms": round(march_launch_ms[0], 4), "l2_gather_peak": L2_GATHER_GBS,
                    "frac_of_l2_gather_peak": round(bytes_a / (march_launch_ms[0] * 1e-3) / 1e9 / L2_GATHER_GBS, 4)},
                "k_ref_shade": {"avg_launch_ms": round(march_launch_ms[2], 4), "traffic": traffic("k_ref_shade<27, true>")}}}
        # warm path (rays resident: the reference's eval semantics, train_eval_pose_est.py:131-149): stage C only, 16 query
        # images per graph against one resident ray set, 4 graphs in flight
        ori, dirs, rgb = pipe.emit(GEN_POINTS, seed=42)
        WQ = 16
        wtok = torch.stack([synthetic.make_tokens(M_TOKENS, 384, seed=100 + q) for q in range(WQ)]).to(device)
        for _ in range(2):
            pipe.identify_batch(wtok, ori, dirs, rgb, TOPK)
        torch.cuda.synchronize(device)
        wgraphs, wstreams = [], [torch.cuda.Stream(device=device) for _ in range(4)]
        for _ in range(4):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                wout = pipe.identify_batch(wtok, ori, dirs, rgb, TOPK)
            wgraphs.append((g, wout))
        torch.cuda.synchronize(device)
        for i in range(8):
            with torch.cuda.stream(wstreams[i % 4]):
                wgraphs[i % 4][0].replay()
        torch.cuda.synchronize(device)
        tw = time.perf_counter()
        n_w = 60
        for i in range(n_w):
            with torch.cuda.stream(wstreams[i % 4]):
                wgraphs[i % 4][0].replay()
        torch.cuda.synchronize(device)
        warm = n_w * WQ / (time.perf_counter() - tw)
        result = {
            "metric": "poses/sec (800x800 query, lego TensoRF)", "value": round(Q * args.steps / dt, 3), "unit": "poses/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "lego-shaped TensorVMSplit %d^3 (16/48 comps, %d^3 mask), gen_points=593 -> 16011 rays, "
                                   "M=256 tokens, top-100, cold path (A+B+C every step)" % (args.grid, max(32, int(round(args.grid * 0.6)))),
                       "queries_per_step": Q, "rays_per_query": GEN_POINTS * 27, "steps_in_flight": in_flight,
                       "emissions_per_step": 1 if sharded else Q,
                       "gemm": "3xBF16 split on the bf16 MFMA (fp32-accurate), fp32 accumulate; march and shading in fp32",
                       "launch": ("hipGraph replay per step (one graph = %d cold queries, each with its own ray set)" % Q)
                                 if not sharded else launch_mode,
                       "parallelism": "single GPU" if not sharded else f"rays sharded over {world_size} ranks + 2 all_gathers"},
            "warm_poses_per_s": round(warm, 2),
            "warm_note": "rays resident (the reference's eval semantics): 16 query images per graph against one ray set, ray "
                         "encoder evaluated once per batch (SURVEY 8f-2), 4 graphs in flight; never part of `value`",
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": roofline,
        }
        if world_size == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ck, idw, tokens[0].cpu())
        else:
            result["cpu_baseline"] = None
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
