#!/usr/bin/env python
"""SSN hot-path benchmark (BASELINE.json metric: proposals/s, 9-segment BNInception SSN fwd+bwd).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or under a launcher

Like the reference, where ONE command starts all GPUs (/root/reference/ssn_train.py:67, DataParallel over
`args.gpus`), `python bench.py --gpus N` is self-contained: when no launcher has set WORLD_SIZE, the process
re-executes itself N times (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), one rank
per GPU over RCCL, waits for the ranks, and rank 0 prints the JSON line.  On a box with fewer GPUs than ranks the
control flow can still be exercised with SSN_BENCH_ONE_DEVICE=1 SSN_BENCH_BACKEND=gloo (all ranks share GPU 0; not
a performance configuration).

One "step" = the loop body of /root/reference/ssn_train.py:205-253 on synthetic THUMOS14-shape
data that is already resident in HBM: SSN forward (backbone -> dropout -> STPP -> heads -> row
selection), the three losses, backward, gradient all-reduce (N > 1) and the SGD update.
Workload per GPU = BASELINE.json configs[1]: 4 videos x 8 proposals x 9 segments = 288 RGB frames
of 224x224 (weak scaling: per-GPU work is fixed as N grows).

Rank 0 prints ONE JSON line.  `roofline` is the MFMA roofline of the dominant kernel family (the
planes implicit-GEMM convolutions `conv_pl_kernel` / `conv_pl9_kernel`, 3 f16 MFMA products per multiply,
forward + dgrad launches), measured live with HIP events around every launch of the same K steps (re-run eagerly right after the timed region, because events cannot be
recorded inside a hipGraph replay); `roofline_detail` lists every conv kernel family; `cpu_baseline` times the CPU oracle
(oracle/ssn_oracle.py, torch fp32 on the host cores) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks fails
# (hipIpcGetMemHandle: invalid argument); already exported on the GPU boxes, kept here for any other launcher
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense f16 / bf16 MFMA (no 2:1 sparsity)
X6_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3  # algorithmic fp32 flops through the 2-way f16 split (3 MFMA products)
PMC_SUMMARY = "r6_pmc_summary.json"
FWD_GFLOP_PER_IMAGE = {"RGB": 4.063152128, "Flow": 4.613883904}  # 2 * conv MACs
# conv1 (7x7/2, 64 outputs of 112x112): 2 * Cin * 49 * 64 * 112^2 flop that a dgrad would cost and nobody needs
CONV1_DGRAD_GFLOP_PER_IMAGE = {"RGB": 2 * 3 * 49 * 64 * 112 * 112 / 1e9, "Flow": 2 * 10 * 49 * 64 * 112 * 112 / 1e9}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--videos-per-gpu", type=int, default=4)
    ap.add_argument("--modality", default="RGB", choices=["RGB", "Flow"])
    ap.add_argument("--num-class", type=int, default=20)
    ap.add_argument("--cpu-baseline-videos", type=int, default=4,
                    help="videos in the CPU-oracle sample (default: the full config-2 batch, 4 videos = 32 proposals; "
                         "0 disables the cpu_baseline leg)")
    ap.add_argument("--cpu-baseline-reps", type=int, default=3, help="timed repetitions of the CPU sample (median)")
    ap.add_argument("--bn-mode", default="frozen", choices=["frozen", "partial", "full"],
                    help="ssn_opts.py --bn_mode; 'frozen' is the reference's default and the configuration of the headline metric")
    ap.add_argument("--precision", default="split", choices=["split", "f32"],
                    help="matrix path of the 1x1/3x3 convolutions: per-tensor scaled 2-way f16 split, 3 products on the "
                         "f16 MFMA (fp32-class error, default) or the exact-f32 MFMA for every layer")
    ap.add_argument("--collectives", default="separate", choices=["separate", "overlapped"],
                    help="N > 1: 'separate' = graph(fwd+bwd) -> eager RCCL all-reduce of the flat gradient -> graph(SGD), "
                         "the well-trodden path; 'overlapped' = bucketed all-reduces issued from inside the backward and "
                         "captured in the one step graph (overlaps the ~0.5 ms ring with the remaining backward)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP event timing")
    ap.add_argument("--single-stream", action="store_true",
                    help="profiling aid: no side streams (weight-gradient chain, forward branches), so that with --no-graph "
                         "every kernel runs alone and rocprofv3's per-kernel durations are free of mutual slow-down")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly from Python instead of replaying one captured hipGraph per step")
    ap.add_argument("--frame-size", type=int, default=0, help=argparse.SUPPRESS)      # tests: emulator mode only
    ap.add_argument("--arch", default="BNInception", choices=["BNInception", "InceptionV3"],
                    help="backbone: BNInception (the headline metric, BASELINE.json configs[1-3]) or InceptionV3 (299x299; the "
                         "reference trains / tests on it with --arch InceptionV3, /root/reference/ssn_models.py:133-139)")
    ap.add_argument("--mode", default="train", choices=["train", "dense-test"],
                    help="train: the loop body of ssn_train.py:205-253 (default); dense-test: the per-video loop of "
                         "ssn_test.py:66-92 (BASELINE.json configs[4]: every sampled frame x 10 crops through the backbone, "
                         "re-organised STPP over the proposals), N > 1 = independent replicas")
    ap.add_argument("--ticks", type=int, default=600, help="dense-test: sampled frames per video")
    ap.add_argument("--proposals", type=int, default=50, help="dense-test: proposals per video")
    ap.add_argument("--tick-batch", type=int, default=60, help="dense-test: ticks per backbone call (the reference uses 4)")
    ap.add_argument("--proposal-list", default="",
                    help="dense-test: a processed proposal list (ops/io.py format); every step scores the NEXT video of the list with its own "
                         "frame count (ticks every 6 frames, ssn_dataset.py:393-396) and its own proposals instead of --ticks / --proposals "
                         "random ones.  tests/golden/proposal_list_processed.txt is the committed excerpt of the ActivityNet-1.2 list")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1, default configuration: do not append the short secondary runs (config 3 = Flow training, config 5 = "
                         "Inception-v3 dense testing on the committed ActivityNet excerpt) under `secondary`")
    return ap.parse_args()


def self_launch(n):
    """No launcher around us: start the N ranks ourselves (one process per GPU), as ssn_train.py:67 starts all its GPUs
    from one command.  Children inherit stdout, so rank 0's JSON line is this command's output; the first failing rank
    takes the others down with it."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if rc:
        raise SystemExit(rc)


def run_secondary(extra, timeout):
    """One secondary configuration in its own process (`python bench.py <extra> --no-secondary`): the fields of its JSON line that
    say what ran, how fast, against which roofline and how close to the oracle."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(extra) + ["--no-secondary"]
    t0 = time.perf_counter()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    try:
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"command": "python bench.py " + " ".join(extra), "error": "timeout after %d s" % timeout}
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    if pr.returncode != 0 or not lines:
        return {"command": "python bench.py " + " ".join(extra), "error": "rc %d" % pr.returncode, "stderr_tail": pr.stderr[-600:]}
    d = json.loads(lines[-1])
    keep = {"command": "python bench.py " + " ".join(extra)}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "final_loss", "videos_per_s",
              "parity_max_rel_logits_vs_cpu_oracle", "parity_max_rel_scores_vs_cpu_oracle", "cpu_baseline"):
        if k in d:
            keep[k] = d[k]
    keep["workload"] = d.get("config", {}).get("workload")
    if "roofline" in d:
        keep["roofline"] = {k: d["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches")}
        keep["frac"] = d["roofline"].get("frac")
    if "scale_guard" in d:
        keep["scale_overflows"] = d["scale_guard"].get("scale_overflows")
    keep["wall_s"] = round(time.perf_counter() - t0, 1)
    return keep


def dense_test_main(args, world, rank, local_rank):
    """--mode dense-test: the per-video loop of /root/reference/ssn_test.py:66-92 (BASELINE.json configs[4]) on synthetic
    ActivityNet-1.2-shape videos resident in HBM: `--ticks` sampled frames x 10 crops through the backbone in large tick
    batches, crop mean, folded test_fc, re-organised STPP over `--proposals` proposals, regression de-normalisation.  One "step" =
    one video.  N > 1: independent replicas (the reference's worker queue, ssn_test.py:145-159), no collective in the data path."""
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    if os.environ.get("SSN_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("SSN_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import numpy as np
    import action_detection_amd as pkg
    from action_detection_amd.dense_test import DenseTester
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic
    pkg.build()
    num_class, crops = 100, 10
    torch.manual_seed(0)
    net = SSN(num_class, 2, 5, 2, "RGB", base_model=args.arch, test_mode=True, stpp_cfg=(1, 1, 1))
    size = net.input_size
    gflop_per_frame = {"BNInception": 4.063152128, "InceptionV3": 2 * 5.711168096}[args.arch]   # 2 * conv MACs
    init_backbone_synthetic(net.base_model)
    init_heads_synthetic(net, std=0.01)
    net.prepare_test_fc()
    net.to(dev).eval()
    tester = DenseTester(net, num_class, stats=np.array([[0.0, 0.0], [1.0, 1.0]]), tick_batch=args.tick_batch)
    g = torch.Generator().manual_seed(1 + rank)
    batch = (torch.randint(0, 256, (crops * args.tick_batch, 3, size, size), generator=g).float() - 110.0).to(dev)
    n_calls = (args.ticks + args.tick_batch - 1) // args.tick_batch
    ticks_total = n_calls * args.tick_batch
    rs = np.random.RandomState(rank)
    starts = rs.randint(0, ticks_total - 8, size=args.proposals)
    lens = rs.randint(2, ticks_total // 3, size=args.proposals)
    pt = np.stack([np.maximum(starts - lens // 2, 0), starts, np.minimum(starts + lens, ticks_total),
                   np.minimum(starts + lens + lens // 2, ticks_total)], axis=1).astype(np.int64)
    sc = rs.rand(args.proposals, 2)
    videos = [(ticks_total, pt, sc)]
    if args.proposal_list:
        # the videos of a processed proposal list, as ssn_dataset.get_test_data prepares them (ssn_dataset.py:393-424): ticks every 6
        # frames, proposal ticks / scaling from the list's own proposals (a video without proposals gets the whole-video one)
        from action_detection_amd.proposal_sampling import ProposalSampler
        sampler = ProposalSampler(prop_file=os.path.join(ROOT, args.proposal_list) if not os.path.isabs(args.proposal_list)
                                  else args.proposal_list, exclude_empty=False, test_interval=6, reg_stats=np.array([[0.0, 0.0], [1.0, 1.0]]))
        videos = []
        for vrec in sampler.video_list:
            vt, _rel, vpt, vsc = sampler.test_ticks(vrec)
            videos.append((len(vt), np.asarray(vpt, dtype=np.int64).reshape(-1, 4), np.asarray(vsc, dtype=np.float64).reshape(-1, 2)))
    batch5 = batch.view(crops, args.tick_batch, 3, size, size)
    part = {}            # remainder batches (a video's last backbone call), crop-major like the full one

    def frames_of(t):
        left = t
        while left > 0:
            b = min(left, args.tick_batch)
            if b == args.tick_batch:
                yield batch
            else:
                if b not in part:
                    part[b] = batch5[:, :b].reshape(crops * b, 3, size, size).contiguous()
                yield part[b]
            left -= b
    vcount = [0]

    def one_video():
        t, vpt, vsc = videos[vcount[0] % len(videos)]
        vcount[0] += 1
        return tester.score_video(frames_of(t), t, torch.from_numpy(vpt), torch.from_numpy(vsc), num_crop=crops), t

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup) * (len(videos) if args.proposal_list else 1)):      # (a list: every video shape once)
        one_video()
    fence()
    vcount[0] = 0
    ticks_done = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ticks_done += one_video()[1]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = ticks_done * crops * world
    list_note = ""
    if args.proposal_list:
        list_note = ("; videos = the %d records of %s in turn (ticks %s, proposals %s)"
                     % (len(videos), args.proposal_list, [v_[0] for v_ in videos], [len(v_[1]) for v_ in videos]))
    result = {
        "metric": "dense-test frames/sec (ssn_test.py per-video loop, %s RGB %dx%d, C=%d)" % (args.arch, size, size, num_class),
        "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "videos_per_s": round(args.steps * world / elapsed, 3),
        "config": {"workload": "%s RGB SSN dense testing, ActivityNet-1.2 shape: %d ticks x %d crops per video (%dx%d), %d proposals, "
                               "backbone in tick batches of %d, crop mean, folded test_fc (out %d), re-organised STPP, de-normalised "
                               "regression; one step = one video, N > 1 = independent replicas"
                               % (args.arch, ticks_done // args.steps, crops, size, size, sum(len(v_[1]) for v_ in videos) // len(videos),
                                  args.tick_batch, net.test_fc.out_features) + list_note,
                   "parallelism": "replicas%d" % world, "layout": net.base_model.layout},
    }
    if rank == 0:
        # roofline of the dominant kernel family: the forward convolutions of one more video, HIP events per launch
        prof = []
        net.base_model.profiler = prof
        overlap, lanes = net.base_model.overlap_wgrad, net.base_model.branch_streams
        net.base_model.branch_streams = False
        vcount[0] = 0
        prof_ticks = one_video()[1]
        torch.cuda.synchronize()
        net.base_model.profiler = None
        net.base_model.branch_streams = lanes
        fl = sum(p[2] for p in prof)
        ms = sum(p[3].elapsed_time(p[4]) for p in prof)
        if prof and ms > 0:
            tf = fl / (ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "kernel": "forward convolution launches (%s)" % "/".join(sorted({p[0] for p in prof})),
                                  "achieved": round(tf, 3), "peak": round(X6_PEAK_TFLOPS, 1), "unit": "TFLOP/s",
                                  "frac": round(tf / X6_PEAK_TFLOPS, 4), "traffic": None,
                                  "peak_note": "algorithmic fp32 flops (2*MACs); peak = 2500 TF dense f16 MFMA / 3 products per multiply",
                                  "launches": len(prof), "conv_ms_per_video": round(ms, 3),
                                  "whole_video_fwd_tflops": round(frames * gflop_per_frame * 1e9 / elapsed / 1e12, 2)}
        if args.cpu_baseline_videos > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ssn_oracle as O
            oracle = O.OracleSSN(num_class, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1), base_model=args.arch)
            oracle.load_state_dict({k: t_.cpu() for k, t_ in net.state_dict().items() if not k.startswith("test_fc")})
            oracle.prepare_test_fc()
            oracle.eval()
            nb = 2
            cb = batch.view(crops, args.tick_batch, 3, size, size)[:, :4].reshape(-1, 3, size, size).cpu()
            cpt = np.minimum(videos[0][1] * (4 * nb) // max(1, videos[0][0]), 4 * nb)      # the first video's proposals on an 8-tick video
            sc_small = videos[0][2]
            c0 = time.perf_counter()
            _oracle_out = O.dense_test_video(oracle, (cb for _ in range(nb)), 4 * nb, cpt, sc_small, num_class, num_crop=crops,
                                             stats=np.array([[0.0, 0.0], [1.0, 1.0]]))
            ct = time.perf_counter() - c0
            # parity in the same run: the product tester on the same 8 ticks x 10 crops and proposals against the oracle's outputs
            ref = _oracle_out
            small = DenseTester(net, num_class, stats=np.array([[0.0, 0.0], [1.0, 1.0]]), tick_batch=4)
            got = small.score_video((cb.to(dev) for _ in range(nb)), 4 * nb, torch.from_numpy(cpt), torch.from_numpy(sc_small), num_crop=crops)
            errs = []
            for g_, r_ in zip(got[:3], ref[:3]):
                if r_ is None:
                    continue
                g_ = g_.float().cpu().numpy()
                # (a range that starts past the last row is a mean over nothing in the reference: NaN on both sides, same places)
                errs.append(float("inf") if not np.array_equal(np.isnan(g_), np.isnan(r_)) else
                            float(np.abs(np.nan_to_num(g_) - np.nan_to_num(r_)).max() / (np.abs(np.nan_to_num(r_)).max() + 1e-20)))
            result["parity_max_rel_scores_vs_cpu_oracle"] = max(errs)
            result["cpu_baseline"] = {"value": round(4 * nb * crops / ct, 2), "unit": "frames/s", "cores": torch.get_num_threads(),
                                      "kind": "port", "sample": "oracle/ssn_oracle.py dense_test_video: %d ticks x %d crops, the "
                                      "reference's batching (4 ticks per call), %.1f s" % (4 * nb, crops, ct)}
        print(json.dumps(result))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher set WORLD_SIZE=%d" % (args.gpus, world))
    if args.mode == "dense-test":
        return dense_test_main(args, world, rank, local_rank)
    # TEST TOOLING (tests/test_bench_selflaunch.py, CPU tier): SSN_BENCH_EMULATOR=1 runs the control flow of this script --
    # self-launch, rendezvous, step loop, collectives, fences, rank-0 JSON -- on the host emulator build of the kernels over gloo,
    # with a stand-in backbone; it measures nothing and says so in the JSON.  Without it there is no CPU path.
    emulator = os.environ.get("SSN_BENCH_EMULATOR") == "1"
    if emulator:
        from action_detection_amd import _lib as _lib_
        _lib_.use_library_for_testing(_lib_.SsnLibrary(os.path.join(ROOT, "tests", "emu", "libssn_emu.so"), is_emulator=True))
        os.environ["SSN_BENCH_BACKEND"] = "gloo"
        args.no_graph = args.no_kernel_events = True
        args.cpu_baseline_videos = 0
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    if os.environ.get("SSN_BENCH_ONE_DEVICE") == "1":   # tooling: several ranks on one GPU (control-flow check with gloo)
        local_rank = 0
    if not emulator and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: this box has %d GPU(s); --gpus %d needs one GPU per rank (SSN_BENCH_ONE_DEVICE=1 "
                         "SSN_BENCH_BACKEND=gloo shares GPU 0 for a control-flow check)"
                         % (rank, torch.cuda.device_count(), args.gpus))
    if emulator:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SSN_FORCE_ALLREDUCE") == "1"
    backend = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("SSN_BENCH_BACKEND", "nccl")   # "nccl" == RCCL; gloo only for the one-GPU control-flow check
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import action_detection_amd as pkg
    from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss, SSNObjective
    from action_detection_amd.optim import SSNSGD
    from action_detection_amd.parallel import GradReducer
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch

    if not emulator:
        pkg.build()
    v = args.videos_per_gpu
    torch.manual_seed(1234 + rank)
    model = SSN(args.num_class, 2, 5, 2, args.modality, dropout=0.8, stpp_cfg=(1, 1, 1), base_model=args.arch, bn_mode=args.bn_mode)
    frame = args.frame_size or model.input_size
    init_backbone_synthetic(model.base_model)  # same weights on every rank (same seed)
    init_heads_synthetic(model, std=0.001)
    if emulator:      # stand-in backbone (pixels -> 1024 features through one trainable matrix): the emulated one takes minutes per frame
        proj = torch.nn.Parameter(torch.randn(3 * 4 * 4, 1024) * 0.01)
        model.base_model.register_parameter("stub_proj", proj)
        model.base_model.features = lambda x: torch.nn.functional.adaptive_avg_pool2d(x, 4).flatten(1) @ proj
    model.base_model.conv_precision = args.precision
    if args.single_stream:
        model.base_model.overlap_wgrad = False
        model.base_model.branch_streams = False
        model.base_model.branch_lanes = False
    model.to(dev).train()
    policies = model.get_optim_policies()
    # (Inception-v3 with these synthetic weights diverges at the reference's default lr = 0.001 -- loss 1e9 after a dozen steps on the
    # fp32 layout, activations past 1e30 after a few more.  Since round 6 the planes path carries such a run on like the reference
    # (non-finite features and gradients, the flagged updates skipped: tests/test_scale_guard.py::test_a_pass_that_overflows_fp32_...),
    # but a line timed on steps that are being redone says nothing: the work per step does not depend on the step size, so it runs at 1e-6)
    opt = SSNSGD(policies, lr=0.001 if args.arch == "BNInception" else 1e-6, momentum=0.9, weight_decay=5e-4)
    overlapped = use_dist and args.collectives == "overlapped"
    reducer = GradReducer(model, deferred=not overlapped) if use_dist else None
    act_crit, comp_crit, reg_crit = ActivityLoss(), CompletenessLoss(), ClassWiseRegressionLoss()
    batch = [t.to(dev) for t in make_batch(v, args.modality, args.num_class, seed=rank, input_size=frame)]
    global_comp_rows = 7 * v * world
    params = [p for g in opt.param_groups for p in g["params"]]
    # Range guard of the planes path's delayed scales (planes_exec.py): inside a graph replay nothing can be polled, so the step
    # carries the range check, the optimizer launches take the fault word and skip a flagged step (weights and momentum stay as
    # they were: the step is retryable), and the host looks at a pinned copy of the word behind every replay.
    fault_word = model.scale_fault_flag(dev)
    # the host reads the word with a FIXED lag of GUARD_LAG steps (a ring of pinned copies, each behind its own event): the copy queued
    # behind step i is looked at behind step i + GUARD_LAG, after its event -- long complete -- has been waited for.  Every rank therefore
    # looks at the word of the SAME step (MAX-reduced over the ranks inside that step), so with N > 1 all ranks redo the same iteration
    # or none does: a rank that redid a step eagerly -- with its collectives -- beside peers replaying their graphs would hang the job.
    GUARD_LAG = 2
    fault_ring = [torch.zeros(1, dtype=torch.int32) for _ in range(GUARD_LAG + 1)]
    if not emulator:
        fault_ring = [t.pin_memory() for t in fault_ring]
    fault_events = [None if emulator else torch.cuda.Event() for _ in fault_ring]
    guard = {"faults": 0, "skipped_steps": 0, "polls": 0, "pending": [], "fault_polls": [], "redone_steps": 0}

    # the objective of ssn_train.py:210-214 (activity CE + 0.1 completeness + 0.1 regression) in one launch each way
    # (ops.ssn_ops.SSNObjective: the three criterions' own loss bodies; SSN_BENCH_SEPARATE_LOSSES=1: the three modules + Python mix)
    objective = SSNObjective(0.1, 0.1)
    seed_one = torch.ones((), device=dev)
    separate_losses = os.environ.get("SSN_BENCH_SEPARATE_LOSSES") == "1"

    def fwd_bwd():
        out = model(*batch)
        if separate_losses:
            loss = (act_crit(out[0], out[1]) + 0.1 * comp_crit(out[2], out[3], 1, 7, global_rows=global_comp_rows)
                    + 0.1 * reg_crit(out[4], out[5], out[6]))
            loss.backward()
            return loss
        loss = objective(*out, sample_split=1, sample_group_size=7, global_rows=global_comp_rows)
        loss.backward(seed_one)
        return loss

    def allreduce_grads():
        """'separate' mode: RCCL all-reduce (sum) of the backbone's flat gradient buffer IN PLACE (the parameter gradients are
        views of it) plus one small bucket for the three heads; the 1/world lands in the SGD kernel."""
        reducer.reduce_all(average=False)

    def update():
        opt.step(grad_scale=(1.0 / world) if (use_dist and not overlapped) else 1.0, skip_flag=fault_word)

    def agree_fault():
        """N > 1: a step is skipped / repeated by ALL ranks or by none (MAX of the fault word; 4 bytes, capturable)."""
        if use_dist:
            reducer.agree_flag_(fault_word)

    # TEST TOOLING (tests/test_bench_selflaunch.py): SSN_BENCH_INJECT_FAULT="rank:step" raises the fault word on ONE rank in ONE step, as a
    # range fault of that rank's backbone would; everything behind it -- the MAX-reduction, the skipped updates, the lagged poll, the redo --
    # is the product protocol
    inject = tuple(int(x) for x in os.environ.get("SSN_BENCH_INJECT_FAULT", "-1:-1").split(":"))
    step_no = [0]

    def step(collectives=True):
        loss = fwd_bwd()
        if collectives and overlapped:
            reducer.reduce_heads()
        elif collectives and use_dist:
            allreduce_grads()
        if collectives and inject == (rank, step_no[0]):
            fault_word.fill_(1)
        step_no[0] += 1
        if collectives:
            agree_fault()
        update()
        opt.zero_grad(set_to_none=True)
        return loss

    def poll_guard():
        """Behind every step: queue a copy of the fault word into pinned memory and look at the copy queued GUARD_LAG steps ago (no
        stall: that step finished long ago; meanwhile the optimizer skips flagged steps on its own).  On a fault: drain, clear,
        recalibrate and redo the step eagerly -- it calibrates as a first step does."""
        k = guard["polls"] % len(fault_ring)
        guard["polls"] += 1
        fault_ring[k].copy_(fault_word, non_blocking=True)
        if fault_events[k] is not None:
            fault_events[k].record()
        guard["pending"].append(k)
        if len(guard["pending"]) <= GUARD_LAG:
            return
        k0 = guard["pending"].pop(0)
        if fault_events[k0] is not None:
            fault_events[k0].synchronize()
        if int(fault_ring[k0][0]) != 0:
            torch.cuda.synchronize()
            guard["faults"] += 1
            guard["fault_polls"].append(guard["polls"] - 1 - GUARD_LAG)     # (which step's word it was: the same on every rank)
            guard["skipped_steps"] += GUARD_LAG + 1      # (the flagged step and the ones queued behind it: their updates were skipped)
            model.recalibrate_scales()
            for t in fault_ring:
                t.zero_()
            guard["pending"].clear()
            # the word is sticky: the flagged step AND the GUARD_LAG steps queued behind it skipped their updates -- all of them are redone
            # (eagerly; the first one calibrates as a first step does), so the run ends with as many updates as it has steps
            for _ in range(GUARD_LAG + 1):
                step()
                guard["redone_steps"] += 1

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- hipGraphs: the ~330 launches of a step are captured once and replayed, so the GPU is not paced by the
    # Python launch loop.  One graph per step (single GPU, or N > 1 with --collectives overlapped); with separate
    # collectives two graphs (forward + backward, optimizer) around the eager all-reduce.  Falls back to eager
    # launches if capture is not possible.
    launch = "eager"
    run_step = step
    static = {}
    if use_dist and overlapped and backend != "nccl" and not args.no_graph:
        # (a gloo collective on device tensors goes through the host and cannot be captured; a failed capture leaves the stream
        # unusable.  Only the one-GPU control-flow check runs this combination: RCCL collectives are captured as usual)
        args.no_graph = True
        launch = "eager (gloo collectives inside the step cannot be captured)"
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):           # allocator / cache warm-up outside the capture
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # with a process group alive, its watchdog thread polls HIP events at any time: in the default "global"
            # capture mode such a call from another thread aborts the capture ("operation not permitted when stream
            # is capturing"), so captures are thread-local whenever torch.distributed is initialised
            cap_mode = "thread_local" if use_dist else "global"
            if use_dist and not overlapped:
                g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb, capture_error_mode=cap_mode):
                    static["loss"] = fwd_bwd()         # parameter .grad tensors become static buffers of this graph
                with torch.cuda.graph(g_up, pool=g_fb.pool(), capture_error_mode=cap_mode):
                    update()

                def run_step():
                    g_fb.replay()
                    allreduce_grads()
                    agree_fault()
                    g_up.replay()
                    return static["loss"]
                launch = "hipGraph replay (fwd+bwd graph, eager RCCL all-reduce, optimizer graph)"
            else:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode=cap_mode):
                    static["loss"] = step()

                def run_step():
                    graph.replay()
                    return static["loss"]
                launch = "hipGraph replay (1 graph = 1 step)"
        except Exception as e:  # noqa: BLE001 -- report and keep going eagerly
            launch = "eager (graph capture failed: %s)" % (str(e).splitlines()[0][:120])
            run_step = step
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
        poll_guard()

    fence()
    eager_repeats_before = dict(model.base_model.guard_stats())      # (calibration of the eager warm-up steps in front of the capture)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
        poll_guard()
    fence()
    elapsed = time.perf_counter() - t0
    eager_repeats_timed = {k: v - eager_repeats_before.get(k, 0) for k, v in model.base_model.guard_stats().items()}
    if any(int(fault_ring[k][0]) != 0 for k in guard["pending"]):      # (the last GUARD_LAG steps: reported, not redone)
        guard["faults"] += 1
        guard["skipped_steps"] += len(guard["pending"])
    rank_ms = 1e3 * elapsed / args.steps

    # ---- per-launch HIP events for the roofline: the same K steps once more, launched eagerly (events cannot
    # be recorded inside a graph replay; the kernels and their arguments are identical)
    prof = None
    if not args.no_kernel_events and rank == 0:
        prof = []
        hbm_prof = []
        import action_detection_amd.kernels as K_
        K_.HBM_PROFILER = hbm_prof
        model.base_model.profiler = prof
        overlap, lanes = model.base_model.overlap_wgrad, model.base_model.branch_streams
        lanes_pl = model.base_model.branch_lanes
        model.base_model.overlap_wgrad = False   # one kernel at a time, so an event pair times exactly one launch
        model.base_model.branch_streams = False
        model.base_model.branch_lanes = False    # (planes executor: the two branch chains of a block back on one stream)
        # rank 0 only: no collectives in this pass (the other ranks are already waiting at the fence below)
        hook, model.base_model.grad_ready_hook = model.base_model.grad_ready_hook, None
        opt.zero_grad(set_to_none=True)
        for _ in range(args.steps):
            step(collectives=False)
        torch.cuda.synchronize()
        model.base_model.grad_ready_hook = hook
        model.base_model.overlap_wgrad, model.base_model.branch_streams = overlap, lanes
        model.base_model.branch_lanes = lanes_pl
        model.base_model.profiler = None
        K_.HBM_PROFILER = None
    fence()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- N > 1 (or SSN_FORCE_ALLREDUCE=1): make the run self-validating -- who took part, on which device, with which RCCL,
    # every rank's own clock, and what the gradient all-reduce costs on its own (same buffers, after the timed region)
    dist_info = None
    if use_dist:
        def _dev_id():
            if emulator:
                return "cpu:%d" % os.getpid()
            pr = torch.cuda.get_device_properties(dev)
            return str(getattr(pr, "uuid", None) or "%s@%s" % (pr.name, getattr(pr, "pci_bus_id", local_rank)))
        mine = {"rank": rank, "local_rank": local_rank, "device": _dev_id(), "ms_per_step": round(rank_ms, 3),
                "guard_faults": guard["faults"], "guard_fault_steps": list(guard["fault_polls"]), "redone_steps": guard["redone_steps"]}
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, mine)
        ar_ms = None
        if not emulator and reducer is not None:
            flat = reducer._deferred_flat
            if flat is None:
                flat = torch.zeros(sum(p.numel() for p in model.base_model.parameters() if p.requires_grad), device=dev)
            dist.all_reduce(flat)                      # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fence()
            e0.record()
            for _ in range(10):
                dist.all_reduce(flat)
            e1.record()
            torch.cuda.synchronize()
            ar_ms = round(e0.elapsed_time(e1) / 10, 4)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:  # noqa: BLE001
            rccl = None
        dist_info = {"ranks_seen": seen, "distinct_devices": len({x["device"] for x in seen}), "rccl_version": rccl,
                     "allreduce_ms_standalone": ar_ms,
                     "allreduce_bytes": (4 * int(flat.numel()) if (not emulator and reducer is not None) else None)}

    proposals = 8 * v * world * args.steps
    value = proposals / elapsed
    if args.arch == "InceptionV3":      # algorithmic conv work from the manifest (2 * MACs; no data gradient for the first layer)
        from action_detection_amd.inceptionv3_spec import build_manifest as _bm, conv_macs as _cm
        _ops, _shapes = _bm(3, frame)
        _first = next(op for op in _ops if op[0] == "conv")
        fwd_gflop_img = 2.0 * _cm(_ops, _shapes) / 1e9
        conv1_dgrad_gflop_img = 2.0 * _shapes[_first[3]][1] * _shapes[_first[3]][2] * _first[5] * _first[6] * _first[7] * _first[8] / 1e9
    else:
        fwd_gflop_img, conv1_dgrad_gflop_img = FWD_GFLOP_PER_IMAGE[args.modality], CONV1_DGRAD_GFLOP_PER_IMAGE[args.modality]
    result = {
        "metric": "proposals/sec (9-seg %s SSN fwd+bwd)" % args.arch,
        "value": round(value, 3),
        "unit": "proposals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not emulator else "synthetic; HOST EMULATOR CONTROL-FLOW RUN with a stand-in backbone -- not a measurement",
        "config": {"workload": "%s %s SSN, %d videos x 8 proposals x 9 segments per GPU (%dx%d), "
                               "fwd + losses + bwd + SGD, THUMOS14 shape (C=%d, stpp [1,1,1], dropout 0.8)"
                               % (args.arch, args.modality, v, frame, frame, args.num_class),
                   "global_batch_proposals": 8 * v * world, "images_per_gpu": 72 * v,
                   "parallelism": "dp%d" % world, "launch": launch,
                   "collectives": (args.collectives if use_dist else "none"),
                   "ranks": (dist.get_world_size() if use_dist else 1),
                   "backend": ({"nccl": "nccl (RCCL over xGMI)"}.get(backend, backend) if use_dist else None),
                   "layout": model.base_model.layout, "bn_mode": args.bn_mode,
                   "conv_precision": ("fp32 parameters / features / gradients; every convolution multiply = 3 f16-MFMA products of "
                                      "per-tensor power-of-two-scaled 2-way f16 operand splits (22 of 24 significand bits per "
                                      "operand, fp32 accumulation: fp32-class error, tests/test_kernels.py K-sweep), the 7x7 "
                                      "stem through its space-to-depth form on the same kernels"
                                      + ("; activations and their gradients are STORED as the two f16 planes (delayed scales)"
                                         if model.base_model.layout == "planes" else "")
                                      if args.precision == "split" else "exact-f32 MFMA everywhere")},
        "final_loss": float(loss.item()),
        "scale_guard": {"protocol": "range check of the delayed scales captured in the step; optimizer launches skip a flagged step "
                                    "(device fault word); host looks at a pinned copy of the word with a fixed lag of 2 steps and redoes a "
                                    "flagged step (and the 2 queued behind it, whose updates the sticky word also skipped) eagerly after recalibration" + ("; the word is MAX-reduced over the ranks" if use_dist else ""),
                        "scale_overflows": guard["faults"], "steps_with_skipped_update": guard["skipped_steps"], "steps_redone_eagerly": guard["redone_steps"],
                        "effective_updates": args.steps + args.warmup - guard["skipped_steps"] + guard["redone_steps"], "repeated_eager_passes": model.base_model.guard_stats(),
                        "repeated_eager_passes_inside_timed_region": eager_repeats_timed,
                        "repeated_eager_passes_note": "passes the range guard repeated while running EAGERLY: the two un-captured warm-up steps in "
                                                      "front of the graph capture (first steps of a fresh state: scales still settling; eager_fault_log "
                                                      "names the tensors) and the per-launch event pass behind the timed region; the timed region itself "
                                                      "replays the graph, where a fault shows as scale_overflows / steps_with_skipped_update",
                        "eager_fault_log": [[w, [[n_, round(v_, 3)] for n_, v_ in bad[:6]]]
                                            for st_ in model.base_model._planes_states.values() for w, bad in st_.fault_log]},
    }
    if dist_info is not None:
        result["distributed"] = dist_info

    oracle_keepalive = []
    if rank == 0:
        # ---------------- roofline of the dominant kernel family (HIP events, timed region) ----------------
        if prof:
            fam = {}
            for family, lid, flops, s, e in prof:
                ms = s.elapsed_time(e)
                f = fam.setdefault(family, [0.0, 0.0, 0])
                f[0] += flops
                f[1] += ms
                f[2] += 1

            def agg(keys):
                fl = sum(fam[k][0] for k in keys if k in fam)
                ms = sum(fam[k][1] for k in keys if k in fam)
                n = sum(fam[k][2] for k in keys if k in fam)
                return fl, ms, n
            # dominant kernel: conv_x6_kernel (forward + dgrad launches of the 1x1/3x3 layers).  Every fp32
            # multiply is 3 f16 MFMA products, so its matrix-pipe ceiling in ALGORITHMIC flops is f16 dense / 3.
            x6_fl, x6_ms, x6_n = agg(("conv_fwd_x6", "conv_dgrad_x6"))
            pl_fl, pl_ms, pl_n = agg(("conv_fwd_pl", "conv_dgrad_pl"))
            if pl_n:     # planes layout: the operands arrive split (csrc/conv_pl.hip); same arithmetic, same ceiling
                x6_n = pl_n
                dom_name = ("conv_pl_kernel (implicit GEMM on planes tensors: activations / gradients stored as 2 f16 terms by "
                            "their producers, 3 v_mfma_f32_32x32x16_f16 per k16 step, no operand conversion in the K loop; "
                            "fwd + dgrad launches)")
                dom_fl, dom_ms, dom_n, dom_peak = pl_fl, pl_ms, pl_n, X6_PEAK_TFLOPS
                pmc_keys = ("conv_pl_kernel_fwd", "conv_pl_kernel_dgrad", "conv_pl9_kernel_fwd", "conv_pl9_kernel_dgrad")
            elif x6_n:
                dom_name = ("conv_x6_kernel (implicit GEMM, fp32 operands scaled per tensor and split into 2 f16 terms, 3 "
                            "v_mfma_f32_32x32x16_f16 per k16 step; fwd + dgrad launches)")
                dom_fl, dom_ms, dom_n, dom_peak = x6_fl, x6_ms, x6_n, X6_PEAK_TFLOPS
                pmc_keys = ("conv_x6_kernel_fwd", "conv_x6_kernel_dgrad")
            else:   # --precision f32: the exact-f32 MFMA kernel carries everything
                dom_name = "conv_igemm_kernel (f32 MFMA implicit GEMM; fwd + dgrad launches)"
                dom_fl, dom_ms, dom_n = agg(("conv_fwd_f32", "conv_dgrad_f32"))
                dom_peak = F32_MFMA_PEAK_TFLOPS
                pmc_keys = ("conv_igemm_kernel_fwd", "conv_igemm_kernel_dgrad")
            achieved = dom_fl / (dom_ms * 1e-3) / 1e12
            # HBM bytes per launch of the same kernel family from the committed PMC passes (FETCH_SIZE, WRITE_SIZE in
            # separate passes, x1024, FETCH doubled for the 16 B/lane read streams as the guide prescribes for gfx950;
            # tools/pmc_summary.py); null if the summary is absent
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", PMC_SUMMARY)) as f:
                    pm = json.load(f)
                fam_p = [pm[k] for k in pmc_keys if k in pm and "hbm_bytes_per_launch" in pm[k]]
                nl = sum(x["launches_sampled"] for x in fam_p)
                traffic = round(sum(x["hbm_bytes_per_launch"] * x["launches_sampled"] for x in fam_p) / nl)
            except (OSError, KeyError, ValueError, ZeroDivisionError):
                pass
            result["roofline"] = {
                "bound": "mfma", "kernel": dom_name,
                "achieved": round(achieved, 3), "peak": round(dom_peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / dom_peak, 4), "traffic": traffic,
                "peak_note": ("algorithmic fp32 flops (2*MACs); peak = 2500 TF dense f16 MFMA / 3 products per multiply"
                              if x6_n else "exact-f32 MFMA peak"),
                "frac_of_f32_mfma_peak": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                "issued_f16_mfma_tflops": round(3 * achieved, 1) if x6_n else None,
                "traffic_note": "HBM bytes/launch, rocprofv3 PMC (profiles/%s)" % PMC_SUMMARY,
                "avg_launch_us": round(1e3 * dom_ms / dom_n, 2), "launches": dom_n,
                "launches_note": "event pairs = API calls: the data gradient of a stride-2 3x3 layer is ONE call that issues four parity-class "
                                 "kernels (rocprofv3 counts those separately: ~101 conv_pl* kernels per step for 87 calls)",
                "algorithmic_gflop_per_launch": round(dom_fl / dom_n / 1e9, 4),
            }
            det = {}
            for k, f in sorted(fam.items()):
                tf = f[0] / (f[1] * 1e-3) / 1e12
                own = X6_PEAK_TFLOPS if k.endswith(("_x6", "_pl")) else F32_MFMA_PEAK_TFLOPS
                det[k] = {"tflops": round(tf, 3), "frac_of_own_mfma_peak": round(tf / own, 4),
                          "frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                          "ms_per_step": round(f[1] / args.steps, 3), "launches_per_step": f[2] // args.steps}
            for grp in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
                fl, ms, n = agg((grp + "_x6", grp + "_f32", grp + "_pl"))
                if n:
                    det[grp + "_all"] = {"tflops": round(fl / (ms * 1e-3) / 1e12, 3),
                                         "frac_of_f32_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                                         "ms_per_step": round(ms / args.steps, 3)}
            det["conv_ms_per_step"] = round(sum(f[1] for f in fam.values()) / args.steps, 3)
            # fwd + dgrad + wgrad = 3 x forward flops minus the data gradient of conv1, which is never computed
            step_gflop = (3 * fwd_gflop_img - conv1_dgrad_gflop_img) * 72 * v
            det["step_algorithmic_gflop"] = round(step_gflop, 1)
            det["whole_step_tflops"] = round(step_gflop * 1e9 / (elapsed / args.steps) / 1e12, 2)
            det["whole_step_frac_of_f32_mfma_peak"] = round(det["whole_step_tflops"] / F32_MFMA_PEAK_TFLOPS, 4)
            det["whole_step_frac_of_split_peak"] = round(det["whole_step_tflops"] / X6_PEAK_TFLOPS, 4)
            result["roofline_detail"] = det

        # ---------------- HBM-bound kernels of the path (STPP, heads, row selection, losses): achieved GB/s ----------
        if prof is not None and hbm_prof:
            hk = {}
            for name, nbytes, s_, e_, reps in hbm_prof:
                h = hk.setdefault(name, [0, 0.0, 0])
                h[0] += nbytes
                h[1] += s_.elapsed_time(e_) / reps
                h[2] += 1
            out_h = {}
            for name, (nbytes, ms, n) in sorted(hk.items()):
                gbps = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                out_h[name] = {"launches_per_step": round(n / args.steps, 2), "bytes_per_launch": int(nbytes / n),
                               "avg_us": round(1e3 * ms / n, 2), "gbps": round(gbps, 1),
                               "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 5)}
            tot_b = sum(h[0] for h in hk.values())
            tot_ms = sum(h[1] for h in hk.values())
            result["hbm_kernels"] = {"peak_gbps": HBM_PEAK_GBPS, "note": "algorithmic bytes (operands + results) / HIP-event "
                                     "duration per launch (each launch repeated %d x back to back inside its event pair: a pair around "
                                     "one 2-5 us launch reads its own 15-25 us granularity); these launches move <= 5 MB each and are "
                                     "latency-bound" % K_.HBM_PROFILER_REPEATS,
                                     "total_ms_per_step": round(tot_ms / args.steps, 4),
                                     "total_gbps": round(tot_b / (tot_ms * 1e-3) / 1e9, 1) if tot_ms else None,
                                     "kernels": out_h}

        # ---------------- CPU baseline: the oracle on the host cores, bounded sample ----------------
        if args.cpu_baseline_videos > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ssn_oracle as O
            cv = args.cpu_baseline_videos
            oracle = O.OracleSSN(args.num_class, 2, 5, 2, args.modality, dropout=0.8, stpp_cfg=(1, 1, 1), base_model=args.arch,
                                 bn_mode=args.bn_mode)
            sd = {k: t.detach().cpu() for k, t in model.state_dict().items()}
            oracle.load_state_dict(sd)
            oracle.train()
            # the identical tensors the GPU was timed on (rank 0's batch) when the sample is the full per-GPU batch
            cb = ([t.detach().cpu() for t in batch] if cv == v
                  else make_batch(cv, args.modality, args.num_class, seed=10_000, input_size=frame))
            times = []
            for rep in range(1 + max(1, args.cpu_baseline_reps)):     # first repetition = warm-up (allocator, threads)
                c0 = time.perf_counter()
                ref = oracle(*cb)
                tot, _, _, _ = O.ssn_total_loss(ref, cv)
                tot.backward()
                times.append(time.perf_counter() - c0)
                oracle.zero_grad(set_to_none=True)
            times = sorted(times[1:])
            ct = times[len(times) // 2]
            result["cpu_baseline"] = {
                "value": round(8 * cv / ct, 4), "unit": "proposals/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": "oracle/ssn_oracle.py (torch-CPU fp32 restatement of the reference SSN), %d videos = %d "
                          "proposals = %d frames%s, fwd + losses + bwd, median of %d after 1 warm-up (%.2f s); the "
                          "reference's own classes (kind 'reference') need /root/reference, which does not exist on the "
                          "GPU box -- the oracle is pinned to them by tests/golden"
                          % (cv, 8 * cv, 72 * cv, " (the tensors of the timed GPU batch)" if cv == v else "",
                             len(times), ct),
            }
            # parity in the same run (eval mode: dropout off on both sides)
            model.eval()
            oracle.eval()
            with torch.no_grad():
                g = model(*[t.to(dev) for t in cb])
                r = oracle(*cb)
            model.train()
            rel = max(((a.float().cpu() - b.float()).abs().max() / (b.float().abs().max() + 1e-20)).item()
                      for a, b in zip(g[0::2], r[0::2]))
            result["parity_max_rel_logits_vs_cpu_oracle"] = rel
        # ---------------- secondary runs (N = 1, default configuration only): BASELINE.json configs[2] and configs[4] ----------------
        # The headline line above is config 2 (configs[1]); the driver runs this one command, so the other two single-GPU
        # configurations are measured here, each in its own short process (own calibration, own graph, own parity check against the
        # oracle) AFTER the timed region, and reported under `secondary` -- never mixed into `value`.
        if (world == 1 and not use_dist and not emulator and not args.no_secondary and args.arch == "BNInception"
                and args.modality == "RGB" and args.bn_mode == "frozen" and args.precision == "split" and not args.no_graph):
            del oracle_keepalive[:]
            torch.cuda.empty_cache()
            result["secondary"] = {
                "flow": run_secondary(["--modality", "Flow", "--steps", "10", "--warmup", "3", "--cpu-baseline-videos", "1",
                                       "--cpu-baseline-reps", "1"], 420),
                "dense_inceptionv3": run_secondary(["--mode", "dense-test", "--arch", "InceptionV3", "--steps", "7", "--warmup", "1",
                                                    "--proposal-list", "tests/golden/proposal_list_processed.txt",
                                                    "--cpu-baseline-videos", "1"], 420),
            }
        print(json.dumps(result))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
