#!/usr/bin/env python
"""bench.py -- ALS rows solved / second over one full iteration (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|small]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full ALS iteration (X-from-Y half + Y-from-X half, Gramians and -- for N>1 -- the
all-gathers and k x k all-reduces included) on the synthetic C4 problem of BASELINE.md
(10M users x 1M items, 1e9 interactions requested, k=64; fits one MI355X).  The same total problem
is used at every N (strong scaling): users and items are row-sharded across the ranks.  Inputs are
generated on the GPU and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (users, items, requested nnz, k, description)
    "c4": (10_000_000, 1_000_000, 1_000_000_000, 64, "C4 synthetic 10M x 1M, 1e9 interactions requested, k=64"),
    "c2": (162_541, 59_047, 25_000_095, 50, "C2 MovieLens-25M shape 162541 x 59047, 25M interactions requested, k=50"),
    "c3": (480_189, 17_770, 100_480_507, 100, "C3 Netflix shape 480189 x 17770, 100M interactions requested, k=100"),
    "c4shard8": (1_250_000, 1_000_000, 125_000_000, 64, "one rank's user rows of C4 at 8 GPUs (1.25M x 1M, 125M interactions requested), k=64"),
    "c5shard8": (12_500_000, 10_000_000, 625_000_000, 128, "one rank's user rows of C5 at 8 GPUs (12.5M x 10M, 625M interactions requested), k=128"),
    # one rank of C5 at 8 GPUs at the shape it really has: BOTH slices of the rank, each gathering from the FULL opposite
    # replica -- 12.5M user rows (625M entries) against the 10M x 128 Y, 1.25M item rows (625M entries, ~500 per row)
    # against the 100M x 128 X (51.2 GB).  (users, items) below = the rows this rank solves per iteration.
    "c5rank": (12_500_000, 1_250_000, 1_250_000_000, 128, "one rank of C5 at 8 GPUs, both slices at true shape: 12.5M user rows x 10M items (625M entries) "
               "+ 1.25M item rows x 100M users (625M entries) gathering from the full 100M x 128 X replica (51.2 GB), k=128"),
    # the same for C4 (the headline configuration at 8 GPUs): 1.25M user rows against the 1M x 64 Y, 125K item rows (~1000 entries
    # each) against the full 10M x 64 X -- what one rank computes per iteration when the exchange is free
    "c4rank": (1_250_000, 125_000, 250_000_000, 64, "one rank of C4 at 8 GPUs, both slices at true shape: 1.25M user rows x 1M items (125M entries) "
               "+ 125K item rows x 10M users (125M entries) gathering from the full 10M x 64 X replica (2.56 GB), k=64"),
    # C5 WHOLE on one device (100M x 10M, 5e9 entries, k=128): ~80 GB of CSR + CSC, 56 GB of factors -- fits
    # the 288 GB of one MI355X.  The N = 1 anchor of a C5 scaling curve and the first handle with more than 2^31 entries.
    # Generated range by range (synth.torch_problem_sliced: one-shot generation does not fit next to the problem).
    "c5": (100_000_000, 10_000_000, 5_000_000_000, 128, "C5 synthetic 100M x 10M, 5e9 interactions, k=128, WHOLE on one GPU"),
    "k128long": (1_000_000, 100_000, 400_000_000, 128, "k=128 with long rows (1M x 100K, 400M interactions requested)"),
    "k112": (2_000_000, 200_000, 200_000_000, 112, "k=112 (2M x 200K, 200M interactions requested)"),
    "k30": (10_000_000, 1_000_000, 1_000_000_000, 30, "the reference's default feature count on the C4 shape (10M x 1M, 1e9 interactions requested, k=30)"),
    "mall400k": (400_000, 100_000, 400_000_000, 64, "cache probe: 400K x 100K, 400M interactions requested, k=64 (X = 102 MB)"),
    "mall2m": (2_000_000, 100_000, 400_000_000, 64, "cache probe: 2M x 100K, 400M interactions requested, k=64 (X = 512 MB)"),
    "small": (200_000, 50_000, 10_000_000, 64, "small smoke workload 200K x 50K, 10M interactions requested, k=64"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(torch, pkg, prob, X, Y, n_users, n_items, k, min_seconds=2.0, units_per_thread=16, max_seconds=12.0):
    """Reference's CPU path, restated (oracle = "port"), timed on this host's cores on a bounded random sample of rows
    and extrapolated by row count.  The sample is sized by TIME, not by count: per side at least `units_per_thread` of the
    reference's 100-row work units (ALS:77, ALS:398-408) for every thread -- so that no thread of a 256-thread host sits
    idle, as it did with a fixed 10 000-row sample -- doubled until the side's solve has run `min_seconds` with that many
    units (or every row is in it, or a run has taken `max_seconds`: the whole baseline stays within ~30 s of CPU work).  The Gramian (serial in the reference, ALS:342 -> MU:219-239) is timed on a row sample too."""
    import numpy as np
    from oracle import oracle
    threads = os.cpu_count() or 1
    rng = np.random.default_rng(1234567890)
    total_t = 0.0
    sample_desc = []
    for side, csr, M, n_rows in ((0, prob["r_csr"], Y, n_users), (1, prob["c_csr"], X, n_items)):
        Mh = M.cpu().numpy()
        g_rows = min(Mh.shape[0], 200000)
        t0 = time.perf_counter()
        G = oracle.gramian(Mh[:g_rows])
        t_g = (time.perf_counter() - t0) * (Mh.shape[0] / g_rows)
        G = G * (Mh.shape[0] / g_rows)  # keep the systems well-posed for the timing run
        rp = csr[0]
        perm = rng.permutation(n_rows)
        # first sample: ~40M entries' worth of rows, never fewer than 4 work units per thread, never more than the
        # `units_per_thread` the loop below aims for; then doubled until it has BOTH run min_seconds and given every thread
        # its units -- or has taken max_seconds (long item rows: 16 units x 256 threads x 1000 entries would be a minute)
        avg_len = max(1.0, float(int(rp[-1])) / max(n_rows, 1))
        target = units_per_thread * 100 * threads                 # what the loop aims for: every thread its units
        floor_units = 4 if avg_len <= 500 else 2                  # long rows (C4's items: ~1000 entries) cost ~6 s per unit per thread
        n_s = int(min(n_rows, max(floor_units * 100 * threads, min(target, 40e6 / avg_len))))
        if n_s >= 0.9 * target:
            n_s = min(n_rows, target)
        while True:
            rows = torch.from_numpy(np.sort(perm[:n_s])).to(rp.device)
            lens = rp[rows + 1] - rp[rows]
            sub_rp = torch.zeros(n_s + 1, dtype=torch.int64, device=rp.device)
            torch.cumsum(lens, 0, out=sub_rp[1:])
            # expand entry indices of the sampled rows
            ent = torch.repeat_interleave(rp[rows] - sub_rp[:-1], lens) + torch.arange(int(sub_rp[-1]), device=rp.device)
            sub_col = csr[1][ent].cpu().numpy()
            sub_val = csr[2][ent].cpu().numpy()
            sub_rp_h = sub_rp.cpu().numpy()
            del ent, lens, rows, sub_rp
            t0 = time.perf_counter()
            oracle.solve_rows(sub_rp_h, sub_col, sub_val, Mh, G, threads=threads)
            t_run = time.perf_counter() - t0
            enough = t_run >= min_seconds and n_s >= min(n_rows, target)
            if enough or t_run >= max_seconds or n_s >= n_rows or len(sub_col) > 600_000_000:
                break
            n_s = min(n_rows, 2 * n_s)
        t_s = t_run * (n_rows / n_s)
        total_t += t_g + t_s
        sample_desc.append("%d of %d %s rows (%d entries, %.1f work units of 100 rows per thread, %.2f s) + Gramian on %d of %d rows" %
                           (n_s, n_rows, "user" if side == 0 else "item", len(sub_col), n_s / 100.0 / threads, t_run, g_rows, Mh.shape[0]))
    return {"value": (n_users + n_items) / total_t, "unit": "rows/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
            "sample": "; ".join(sample_desc) + "; per-side times extrapolated by row count; Gramian single-threaded as in the reference"}


def jvm_baseline(torch, prob, n_users, n_items, k, sample_users=20000):
    """Opportunistic "reference JVM path" (BASELINE.md section 4): only when java, javac and the reference's
    jars (env MYRRIX_CP) exist on this host -- never in the build image.  Times the real
    AlternatingLeastSquares through java/bench/ReferenceAlsTimer.java (ours) on the rows of a user sample
    restricted to the items they touch.  Returns None when the toolchain is absent."""
    import shutil
    import struct
    import subprocess
    import tempfile
    cp = os.environ.get("MYRRIX_CP")
    if not cp or not shutil.which("java") or not shutil.which("javac"):
        return None
    import numpy as np
    rp, col, val = prob["r_csr"]
    users = np.sort(np.random.default_rng(99).choice(n_users, size=min(sample_users, n_users), replace=False))
    rp_h = rp.cpu().numpy()
    tmp = tempfile.mkdtemp(prefix="mals_jvm_")
    path = os.path.join(tmp, "entries.bin")
    n = 0
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 0))
        for u in users:
            a, b = int(rp_h[u]), int(rp_h[u + 1])
            cs, vs = col[a:b].cpu().numpy(), val[a:b].cpu().numpy()
            for c, v in zip(cs, vs):
                f.write(struct.pack("<qqf", int(u), int(c), float(v)))
            n += b - a
        f.seek(0)
        f.write(struct.pack("<i", n))
    src = os.path.join(ROOT, "java", "bench", "ReferenceAlsTimer.java")
    try:
        subprocess.check_call(["javac", "-cp", cp, "-d", tmp, src])
        out = subprocess.check_output(["java", "-cp", cp + os.pathsep + tmp, "bench.ReferenceAlsTimer", path, str(k), "2"],
                                      text=True, timeout=600)
    except (subprocess.SubprocessError, OSError) as e:
        return {"error": str(e)[:200]}
    line = [ln for ln in out.splitlines() if ln.startswith("rows_per_s=")][-1]
    fields = dict(t.split("=") for t in line.split())
    return {"value": float(fields["rows_per_s"]), "unit": "rows/s", "cores": int(fields["threads"]), "kind": "reference",
            "sample": "the real net.myrrix AlternatingLeastSquares on %d sampled users x the items they touch (%d entries), JVM on this host" % (len(users), n)}


def rank_problem(torch, synth, n_users, n_items, nnz_side, k, device, world_emulated=8, rank=3, planted=0.3):
    """Both slices of ONE rank of a `world_emulated`-GPU run, each against the FULL opposite side (the rank's shape of
    SURVEY.md 8(e) / App. C): user rows [u_off, u_off + n_users) of R over all n_items * world items, item rows
    [i_off, i_off + n_items) of R^T over all n_users * world users."""
    n_users_total, n_items_total = n_users * world_emulated, n_items * world_emulated
    p = synth.torch_problem(n_users, n_items_total, nnz_side, k, device, planted=planted)
    r_csr, Y0, planted_desc = p["r_csr"], p["Y0"], p["planted"]
    del p
    torch.cuda.empty_cache()
    c_csr = synth.torch_slice(n_items, n_users_total, nnz_side, device, rows="items")
    g = torch.Generator(device=device)
    g.manual_seed(synth.SEED + 2)
    # the X replica as the other ranks' all-gathers would have left it: rows of the size a solved user row has
    X0 = torch.randn(n_users_total, k, generator=g, device=device, dtype=torch.float32)
    X0 *= 0.3 / k ** 0.5
    return {"r_csr": r_csr, "c_csr": c_csr, "Y0": Y0, "X0": X0, "nnz": int(r_csr[1].numel() + c_csr[1].numel()), "planted": None,
            "planted_user_slice": planted_desc, "rank": rank, "u_off": rank * n_users, "i_off": rank * n_items,
            "n_users_total": n_users_total, "n_items_total": n_items_total}


class RankDriver:
    """One rank of a group without its peers: the rank's own slices are solved for real, the rows of the other ranks
    stay what they were (as if their all-gathers had delivered them), and the shared Gramian is the rank's partial over
    its own 1/world of the rows + the (constant) sum of the others' partials -- what the k x k all-reduce would deliver.
    No exchange is timed: this prices the COMPUTE of a C5 rank at its true shape."""

    def __init__(self, torch, pkg, core, prob, k, device):
        self.torch, self.pkg, self.core, self.k = torch, pkg, core, k
        self.off = {pkg.SIDE_X: prob["u_off"], pkg.SIDE_Y: prob["i_off"]}
        self.n = {pkg.SIDE_X: prob["r_csr"][0].numel() - 1, pkg.SIDE_Y: prob["c_csr"][0].numel() - 1}
        self.tot = {pkg.SIDE_X: prob["n_users_total"], pkg.SIDE_Y: prob["n_items_total"]}
        self.F = {pkg.SIDE_X: prob["X0"], pkg.SIDE_Y: torch.zeros(self.tot[pkg.SIDE_Y], k, dtype=torch.float32, device=device)}
        self.F[pkg.SIDE_Y][:prob["Y0"].shape[0]].copy_(prob["Y0"])
        for side in (pkg.SIDE_X, pkg.SIDE_Y):
            core.bind_factors(side, self.F[side])
        core.set_matrix(pkg.SIDE_X, *prob["r_csr"], row_offset=self.off[pkg.SIDE_X])
        core.set_matrix(pkg.SIDE_Y, *prob["c_csr"], row_offset=self.off[pkg.SIDE_Y])
        self.per = self.n
        self._gp = torch.zeros(k, k, dtype=torch.float64, device=device)
        self._g = torch.zeros(k, k, dtype=torch.float64, device=device)
        self.rest = {}
        for side in (pkg.SIDE_X, pkg.SIDE_Y):      # the other ranks' partial Gramians: constant here
            acc = torch.zeros(k, k, dtype=torch.float64, device=device)
            for lo, hi in ((0, self.off[side]), (self.off[side] + self.n[side], self.tot[side])):
                if hi > lo:
                    core.gramian_partial(side, lo, hi - lo, self._gp)
                    torch.cuda.synchronize()
                    acc += self._gp
            self.rest[side] = acc

    def _gramian(self, side):
        self.core.gramian_partial(side, self.off[side], self.n[side], self._gp)
        self.torch.add(self._gp, self.rest[side], out=self._g)
        self.core.set_gramian(side, self._g)

    def half_iteration(self, side):
        self._gramian(1 - side)
        self.core.solve_side(side)

    def iterate(self, n=1, check=True):
        for _ in range(n):
            for side in (self.pkg.SIDE_X, self.pkg.SIDE_Y):
                self.half_iteration(side)
                if check:
                    self.core.check()

    def _all_gather(self, side):
        pass

    def factors(self, side):
        return self.F[side]


def unplanted_leg(torch, pkg, sharded, synth, n_users, n_items, nnz_req, k, args, gmode, smode, local_rank, device, prob=None, label=None):
    """The same measurement on the plain SURVEY 8(d) workload (power-law item popularity, log-normal user activity, values
    1..5, NOTHING planted), beside the headline: the planted part makes 30 % of the X-half's gathers hit 2048 hot item rows
    and moves ~110M entries onto the long-row kernel.  One GPU, same build, same steps.
    With `prob` given: the same measurement on THAT problem under another arithmetic mode (roofline_fp32)."""
    own = prob is None
    if own:
        prob = synth.torch_problem(n_users, n_items, nnz_req, k, device, planted=0.0)
    core = pkg.ALSCore(k, alpha=1.0, lam=0.1, device=local_rank, segment_nnz=args.segment_nnz, gramian_mode=gmode, solve_mode=smode)
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    als = sharded.ShardedALS(core, n_users, n_items, k, rank=0, world=1, device=device)
    als.set_matrix_from_full(pkg.SIDE_X, *prob["r_csr"])
    als.set_matrix_from_full(pkg.SIDE_Y, *prob["c_csr"])
    als.set_factors(pkg.SIDE_Y, prob["Y0"])
    core.enable_timing(True)
    als.iterate(args.warmup, check=True)
    torch.cuda.synchronize()
    warm = core.stats()
    core.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    als.iterate(args.steps, check=False)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    core.check()
    st = core.stats()
    core.enable_timing(False)
    n_launch = max(st["rows_launches"], 1)
    avg_ms = st["rows_ms"] / n_launch
    achieved = st["rows_bytes"] / n_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    it_bytes = (st["rows_bytes"] + st["segments_bytes"] + st["finish_bytes"] + st["dual_bytes"] + st["gramian_bytes"]) / args.steps
    ms = 1e3 * elapsed / args.steps
    out = {"workload": label or "the same shape with nothing planted (SURVEY.md 8(d) as written)", "nnz": int(prob["nnz"]),
           "ms_per_step": ms, "value": (n_users + n_items) / (elapsed / args.steps), "unit": "rows/s",
           "kernel": "rows kernel (als_persistent_kernel_h, MODE 0)", "avg_launch_ms": avg_ms, "achieved": achieved, "peak": HBM_PEAK_GBS,
           "frac": achieved / HBM_PEAK_GBS, "iteration_frac": it_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "kernels_ms_per_step": {name: st[name + "_ms"] / args.steps for name in ("rows", "segments", "finish", "gramian", "dual", "rotate")},
           "rows_refined_per_step": st["rows_refined"] / args.steps,
           "_all_rows_launches": (warm["rows_launches"] + st["rows_launches"], warm["rows_ms"] + st["rows_ms"])}
    out["gather_scale"] = core.gather_scale() if gmode != 1 else None
    core.close()
    del als
    if own:
        del prob
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--segment-nnz", type=int, default=0)
    ap.add_argument("--gramian-mode", default="auto", choices=["auto", "fp32", "split_f16", "split3_f16"],
                    help="mals_config.gramian_mode (A/B only; the headline number uses the library default)")
    ap.add_argument("--solve-mode", default="auto", choices=["auto", "direct", "dual"],
                    help="mals_config.solve_mode (A/B only; the headline number uses the library default)")
    ap.add_argument("--exchange", default="group", choices=["group", "torch"],
                    help="N>1: 'group' = the library's own RCCL exchange below the C-ABI (mals_group_*, cost-balanced slices); "
                         "'torch' = equal-row slices exchanged with torch.distributed collectives (sharded.py)")
    ap.add_argument("--planted", type=float, default=0.3,
                    help="fraction of the interactions planted on the 2048 core items (synth.torch_problem); 0 = the plain SURVEY 8(d) workload "
                         "(power-law item popularity, log-normal user activity, values 1..5, nothing planted)")
    ap.add_argument("--no-unplanted", action="store_true",
                    help="N=1: skip the second, untimed-by-the-headline leg on the un-planted workload (roofline_unplanted)")
    ap.add_argument("--one-compute-stream", action="store_true",
                    help="group path: solve every chunk on ONE compute stream (A/B against the default, two alternating streams)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="N=1: skip the leg that re-runs the workload with --gramian-mode fp32 (roofline_fp32)")
    ap.add_argument("--exchange-chunks", type=int, default=4,
                    help="N>1: solve each slice in this many row chunks and all-gather a finished chunk while the next is solved")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU over RCCL
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execvp(cmd[0], cmd)

    import torch
    import torch.distributed as dist
    import myrrix_recommender_amd as pkg
    from myrrix_recommender_amd import sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MALS_BENCH_ONE_DEVICE=1 (tests only, tests/test_gpu_group_transport.py): every rank on device 0 and torch's own
    # rendezvous on gloo -- with MALS_BENCH_TRANSPORT pointing at the tests' stand-in transport (this SCRIPT then calls
    # mals_group_use_transport; the library itself reads no such variable) this runs the whole N > 1 flow of this script
    # on a box with one GPU.  The numbers of such a run mean nothing.
    one_device = os.environ.get("MALS_BENCH_ONE_DEVICE", "0") == "1"
    transport = os.environ.get("MALS_BENCH_TRANSPORT")
    if transport:
        pkg.GroupALS.use_transport(transport)
    if one_device:
        local_rank = 0
    if world != args.gpus:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MALS_FORCE_COLLECTIVES=1 (diagnostic): run the multi-GPU code path -- chunked solves, partial
    # Gramian + all-reduce, all-gathers -- with a single rank, to price everything but the wire
    force = os.environ.get("MALS_FORCE_COLLECTIVES", "0") == "1"
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    red_device = torch.device("cpu") if one_device else device   # where the few scalars of this script are reduced

    n_users, n_items, nnz_req, k, desc = WORKLOADS[args.workload]
    rank_shape = args.workload in ("c5rank", "c4rank")
    if rank_shape:
        assert world == 1 and not force, "c5rank / c4rank emulate ONE rank of an 8-GPU group on one GPU"
    t_gen = time.perf_counter()
    if rank_shape:
        prob = rank_problem(torch, synth, n_users, n_items, nnz_req // 2, k, device, world_emulated=8, planted=args.planted)
    elif args.workload == "c5":
        assert world == 1, "the c5 workload is the whole problem on ONE device"
        prob = synth.torch_problem_sliced(n_users, n_items, nnz_req, k, device, slices=8)
    else:
        prob = synth.torch_problem(n_users, n_items, nnz_req, k, device, planted=args.planted)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    chunk_rows = 0
    gmode = {"auto": 0, "fp32": 1, "split_f16": 2, "split3_f16": 3}[args.gramian_mode]
    smode = {"auto": 0, "direct": 1, "dual": 2}[args.solve_mode]
    use_group = (world > 1 or force) and args.exchange == "group"
    if rank_shape:
        core = pkg.ALSCore(k, alpha=1.0, lam=0.1, device=local_rank, segment_nnz=args.segment_nnz, gramian_mode=gmode, solve_mode=smode)
        core.set_stream(torch.cuda.current_stream().cuda_stream)
        als = RankDriver(torch, pkg, core, prob, k, device)
        slice_info = {"emulated_rank": prob["rank"], "emulated_world": 8, "x_rows": [prob["u_off"], prob["u_off"] + n_users], "y_rows": [prob["i_off"], prob["i_off"] + n_items],
                      "replica_rows": {"x": prob["n_users_total"], "y": prob["n_items_total"]},
                      "hbm_GB_after_setup": round(torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9, 1)}
    elif use_group:
        # one rank per process; torch.distributed only carried the RCCL unique id (and the barriers of the
        # timing bracket): slices, partial Gramians, all-reduce and the chunked exchange are the library's
        grp = pkg.GroupALS.from_torch_distributed(k, local_rank, alpha=1.0, lam=0.1, segment_nnz=args.segment_nnz, gramian_mode=gmode,
                                                  solve_mode=smode, exchange_chunks=args.exchange_chunks, world=world, rank=rank,
                                                  one_rank_communicator=force)
        if args.one_compute_stream:
            grp.set_alternate_streams(False)
        grp.set_factor_rows(pkg.SIDE_X, n_users)
        grp.set_factor_rows(pkg.SIDE_Y, n_items)
        grp.set_matrix(pkg.SIDE_X, *prob["r_csr"])
        grp.set_matrix(pkg.SIDE_Y, *prob["c_csr"])
        grp.set_factors(pkg.SIDE_Y, prob["Y0"].cpu().numpy())
        core = grp.local(0)[0]

        class GroupDriver:
            per = {pkg.SIDE_X: (n_users + world - 1) // world, pkg.SIDE_Y: (n_items + world - 1) // world}

            def iterate(self, n, check=True):
                grp.iterate(n)          # every half-iteration ends with the agreed status (ALS:346-361)

            def half_iteration(self, side):
                grp.half_iteration(side)

            def _all_gather(self, side):
                grp.exchange_only(side)

            def factors(self, side):
                ptr, n = core.factor_device_ptr(side)

                class _View:
                    __cuda_array_interface__ = {"shape": (n, k), "typestr": "<f4", "data": (ptr, False), "version": 2}
                return torch.as_tensor(_View(), device=device)[:(n_users if side == pkg.SIDE_X else n_items)]
        als = GroupDriver()
        slice_info = {"x_bounds": grp.bounds(pkg.SIDE_X).tolist(), "y_bounds": grp.bounds(pkg.SIDE_Y).tolist()}
    else:
        if (world > 1 or force) and args.exchange_chunks > 1:
            upr = sharded.rows_per_rank(n_users, world)
            chunk_rows = (upr + args.exchange_chunks - 1) // args.exchange_chunks
        core = pkg.ALSCore(k, alpha=1.0, lam=0.1, device=local_rank, segment_nnz=args.segment_nnz, chunk_rows=chunk_rows,
                           gramian_mode=gmode, solve_mode=smode)
        core.set_stream(torch.cuda.current_stream().cuda_stream)
        als = sharded.ShardedALS(core, n_users, n_items, k, rank=rank, world=world, device=device, force_collectives=force)
        als.set_matrix_from_full(pkg.SIDE_X, *prob["r_csr"])
        als.set_matrix_from_full(pkg.SIDE_Y, *prob["c_csr"])
        als.set_factors(pkg.SIDE_Y, prob["Y0"])
        slice_info = None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    core.enable_timing(True)     # HIP events around every kernel from the first launch on: the tally over ALL launches of the
    als.iterate(args.warmup, check=True)   # process is what a rocprofv3 --kernel-trace --stats average of this command shows
    torch.cuda.synchronize()
    tally = {"rows": [0, 0.0], "dual": [0, 0.0]}

    def add_tally(stt):
        for kk in tally:
            tally[kk][0] += stt[kk + "_launches"]
            tally[kk][1] += stt[kk + "_ms"]
    add_tally(core.stats())
    core.reset_stats()
    # THE timed region: K steps with the status check of every half-iteration inside (mals_check: one D2H of the bad-row /
    # suspect words + a stream sync per half -- what mals_factorize and the group path always pay; ALS:346-361)
    barrier()
    t0 = time.perf_counter()
    als.iterate(args.steps, check=True)
    barrier()
    elapsed = time.perf_counter() - t0
    st = core.stats()
    add_tally(st)
    # the same K steps again without the per-half check (a diagnostic: what the check costs)
    core.reset_stats()       # (st above keeps the timed region's tallies)
    barrier()
    t0 = time.perf_counter()
    als.iterate(args.steps, check=False)
    barrier()
    elapsed_chk = time.perf_counter() - t0
    core.check()
    add_tally(core.stats())
    core.enable_timing(False)
    gscale = core.gather_scale()
    # untimed diagnostic pass: per-half kernel times (not part of the measured region)
    halves = {}
    if os.environ.get("MALS_BENCH_SPLIT", "1") == "1":
        core.enable_timing(True)
        for side, name in ((pkg.SIDE_X, "x_half"), (pkg.SIDE_Y, "y_half")):
            core.reset_stats()
            als.half_iteration(side)
            torch.cuda.synchronize()
            h = core.stats()
            halves[name] = {kk: round(h[kk + "_ms"], 3) for kk in ("rows", "segments", "finish", "gramian", "dual", "rotate")}
            halves[name]["rows_dual"] = h["rows_dual"]
            halves[name]["rows_GBps"] = round(h["rows_bytes"] / max(h["rows_ms"], 1e-9) / 1e6, 1)
            halves[name]["segments_GBps"] = round(h["segments_bytes"] / max(h["segments_ms"], 1e-9) / 1e6, 1) if h["segments_bytes"] else None
            halves[name]["dual_GBps"] = round(h["dual_bytes"] / max(h["dual_ms"], 1e-9) / 1e6, 1) if h["dual_bytes"] else None
            add_tally(h)
        core.enable_timing(False)
    # untimed: the exchange on its own ("report all-gather time separately") -- the
    # un-pipelined in-place all-gather of each side's freshly solved slices into every replica
    exchange = None
    if world > 1 or force:
        exchange = {}
        for side, name in ((pkg.SIDE_X, "x_ms"), (pkg.SIDE_Y, "y_ms")):
            als._all_gather(side)
            barrier()
            te = time.perf_counter()
            for _ in range(3):
                als._all_gather(side)
            barrier()
            exchange[name] = (time.perf_counter() - te) / 3 * 1e3
        exchange["bytes_received_per_rank"] = {"x": (world - 1) * als.per[pkg.SIDE_X] * k * 4, "y": (world - 1) * als.per[pkg.SIDE_Y] * k * 4}
    # untimed quality figure: ReconstructionEvaluator's mean over the observed entries (8(f) row 3)
    rec_sum, rec_cnt = core.reconstruction_error()
    own_elapsed = elapsed
    if world > 1:
        t = torch.tensor([elapsed, elapsed_chk], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_chk = float(t[0].item()), float(t[1].item())
        q = torch.tensor([rec_sum, float(rec_cnt)], dtype=torch.float64, device=red_device)
        dist.all_reduce(q)
        rec_sum, rec_cnt = float(q[0].item()), int(q[1].item())

    # N > 1: what every rank saw, so that the line proves itself -- size and rank READ BACK from the RCCL communicator,
    # the HIP device and its PCI bus id, the rows / entries of the rank's two slices, the rank's own clock
    ranks_info = None
    if world > 1 or force:
        mine = {"rank": rank, "local_rank": local_rank, "hip_device": torch.cuda.current_device(), "ms_per_step": 1e3 * own_elapsed / args.steps,
                "device_name": torch.cuda.get_device_name(local_rank)}
        if use_group:
            mine.update(grp.comm_info(0))
            for side, name in ((pkg.SIDE_X, "x"), (pkg.SIDE_Y, "y")):
                b = grp.bounds(side)
                rp = prob["r_csr" if side == pkg.SIDE_X else "c_csr"][0]
                mine[name + "_rows"] = int(b[rank + 1] - b[rank])
                mine[name + "_nnz"] = int(rp[int(b[rank + 1])] - rp[int(b[rank])])
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
        else:
            gathered = [mine]
        ranks_info = gathered
    planted_err = None
    if world == 1:
        planted_err = synth.planted_reconstruction_error(prob, als.factors(pkg.SIDE_X), als.factors(pkg.SIDE_Y))
    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # dominant kernel = the one with the most time in the timed region: the fused gather + Gramian +
        # Cholesky kernel over the direct rows (MODE 0; als_persistent_kernel_h is what AUTO selects above
        # k = 32) or the dual kernels of the short rows (als_dual_kernel<T,TN>, all row classes together)
        split = args.gramian_mode == "split_f16" or (args.gramian_mode == "auto" and k > 16)
        T_blocks = (k + 15) // 16
        dom = "dual" if st["dual_ms"] > st["rows_ms"] else "rows"
        if dom == "dual":
            kernel_name = "mals::als_dual_kernel<T=%d,TN=1..%d> (short rows: gather of rotated rows + n_u x n_u system, all row classes of a half-iteration)" % ((k + 15) // 16, (k + 15) // 32)
            # one "launch" = the dual kernels of one half-iteration (up to 4 row classes back to back)
            n_launch = max(st["rows_launches"], 1) if st["rows_launches"] else max(st["dual_launches"], 1)
        else:
            if split and k == 128 and os.environ.get("MALS_LDS_GATHER", "1") != "0":
                kernel_name = "mals::als_lds_kernel_h<MODE=0> (k = 128: gather staged through LDS by global_load_lds, split-f16 Gramian in 32-entry super-steps + Cholesky, rows)"
            else:
                kernel_name = ("mals::als_persistent_kernel_h<T=%d,MODE=0> (fused gather + split-f16 Gramian + Cholesky, rows)"
                               if split else
                               "mals::als_persistent_kernel<T=%d,D,MODE=0> (fused gather + fp32 Gramian + Cholesky, rows)") % ((k + 15) // 16)
            n_launch = max(st["rows_launches"], 1)
        avg_ms = st[dom + "_ms"] / n_launch
        bytes_per_launch = st[dom + "_bytes"] / n_launch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the other roofline of the per-row systems: fp32-equivalent matrix-pipe work of the factorizations
        # (k^3/3 + 2 k^2 flop per direct row) against the dense fp32 MFMA peak
        k3_flop = (st["rows_solved"] - st["rows_dual"]) / args.steps * (k ** 3 / 3.0 + 2.0 * k * k)
        k3_ms_at_peak = k3_flop / 157.3e12 * 1e3
        # HBM traffic per launch of the same kernel from the PMC passes committed under profiles/
        # (rocprofv3 cannot run inside this process); null when no profile matches this workload
        traffic, traffic_src, gram_busy = None, None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pm.get("workload") == args.workload and pm.get("k") == k and world == 1:
                if dom == "rows":
                    traffic = pm["traffic_bytes_per_launch"]
                    traffic_src = "profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
                gram_busy = pm.get("gramian_mfma_busy_fraction")
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "ALS rows solved/sec (full iteration) at k=%d" % k,
            "value": (n_users + n_items) / (elapsed / args.steps),
            "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            # (value / ms_per_step include mals_check after every half-iteration; the same K steps again without it:)
            "ms_per_step_without_check": 1e3 * elapsed_chk / args.steps,
            "value_without_check": (n_users + n_items) / (elapsed_chk / args.steps),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # the arithmetic type of the path: factors are fp32 like the reference's; above 16 features the per-row Gramian
            # takes its operands split into two f16 halves (22 significand bits, exact products) and accumulates in fp32;
            # `roofline_fp32` below is the same workload with fp32 products end to end (--gramian-mode fp32)
            "dtype": ("f32 storage; per-row Gramian: f16x2-split operands (22 bits), f32 accumulate; Cholesky f32" if split else "f32"),
            "data": "synthetic",
            "config": {"workload": desc, "users": n_users, "items": n_items, "nnz": int(prob["nnz"]), "features": k, "planted": prob.get("planted"),
                       "alpha": 1.0, "lambda": 0.1,
                       "arithmetic": ("per-row Gramian: operands split into two f16 halves (22 significand bits), exact products, fp32 accumulate"
                                      if split else "per-row Gramian: fp32 products, fp32 accumulate") +
                                     "; M^T M: fp64; Cholesky and solves: fp32; factors stored fp32 like the reference",
                       "sharding": ("rows x%d, cost-balanced slices, RCCL below the C-ABI (mals_group_*): kxk all-reduce + exchange in %d chunks behind the solve; "
                                    "communicator size read back from every rank: %s"
                                    % (world, args.exchange_chunks, sorted({r.get("comm_size") for r in ranks_info}) if ranks_info else "n/a"))
                                   if use_group else
                                   ("rows x%d, %s all-gather + kxk all-reduce (torch.distributed)" % (world, "in-place" if chunk_rows == 0 else "chunked (%d rows) pipelined" % chunk_rows)),
                       "slices": slice_info,
                       "setup_s": round(t_gen, 2)},
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # the same fraction on the COUNTER basis: HBM bytes the memory system really moved per launch
                         "frac_counter": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms > 0 else None,
                         "iteration_frac": (st["rows_bytes"] + st["segments_bytes"] + st["finish_bytes"] + st["dual_bytes"] + st["gramian_bytes"]) / args.steps / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launches": n_launch,
                         "iteration": {"algorithmic_bytes": (st["rows_bytes"] + st["segments_bytes"] + st["finish_bytes"] + st["dual_bytes"] + st["gramian_bytes"]) / args.steps,
                                       "hbm_floor_ms": (st["rows_bytes"] + st["segments_bytes"] + st["finish_bytes"] + st["dual_bytes"] + st["gramian_bytes"]) / args.steps / (HBM_PEAK_GBS * 1e6),
                                       "direct_factorization_fp32_mfma_floor_ms": k3_ms_at_peak,
                                       "binding": "hbm" if (st["rows_bytes"] + st["segments_bytes"] + st["finish_bytes"] + st["dual_bytes"] + st["gramian_bytes"]) / args.steps / (HBM_PEAK_GBS * 1e6) >= k3_ms_at_peak else "mfma_fp32"}},
            "kernels_ms_per_step": {name: st[name + "_ms"] / args.steps for name in ("rows", "segments", "finish", "gramian", "dual", "rotate")},
            "gramian": {"ms_per_step": st["gramian_ms"] / args.steps,
                        "kernel": "mals::gramian_split_kernel<T=%d> (M^T M on %s, split-f16 operands, fp32 slab sums + fp64 across slabs) "
                                  "from 262144 rows on; mals::gramian_partial_kernel<T=%d> (v_mfma_f64_16x16x4_f64) below"
                                  % (T_blocks, "v_mfma_f32_16x16x32_f16" if T_blocks <= 4 else "v_mfma_f32_16x16x16_f16", T_blocks),
                        # rows x tri(T) upper tiles x 512 flop per row and tile (2 x 16 x 16), whichever kernel ran
                        "effective_TFLOPs": (st["gramian_bytes"] / (4.0 * k)) * (T_blocks * (T_blocks + 1) // 2) * 512.0 / max(st["gramian_ms"], 1e-9) / 1e9,
                        "mfma_busy_frac": gram_busy,
                        "mfma_busy_source": "profiles/pmc_traffic.json (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), the Gramian kernel with the most time)" if gram_busy is not None else None},
            # {S, 1/S^2, range flag, bound on |y|} of the last split-precision gather: flag 1 = the split-f16 kernels ran, 0 = their
            # fp32 twins (c5rank / c4rank install the Gramian through mals_set_gramian: the bound is sqrt(max G_ff) there)
            "gather_scale": gscale,
            "rows_dual_per_step": st["rows_dual"] / args.steps,
            "rows_refined_per_step": st["rows_refined"] / args.steps,   # ill-conditioned rows re-solved with fp64 residuals
            "eigen_host_ms_per_step": st["eigen_host_ms"] / args.steps,
            "half_iteration_kernel_ms": halves,
            "all_gather_alone_ms": exchange,
            "ranks": ({"per_rank": ranks_info,
                       "comm_sizes_read_back": sorted({r.get("comm_size") for r in ranks_info}),
                       "distinct_devices": len({(r.get("pci_bus_id") or r["hip_device"]) for r in ranks_info}),
                       "ms_per_step_min": min(r["ms_per_step"] for r in ranks_info), "ms_per_step_max": max(r["ms_per_step"] for r in ranks_info)}
                      if ranks_info else None),
            "reconstruction_error": {"mean": rec_sum / max(rec_cnt, 1), "entries": rec_cnt,
                                     "what": "mean over stored entries of max(0, 1 - x_u.y_i) after warmup+steps iterations "
                                             "(ReconstructionEvaluator.java:91-102), untimed; planted_part = the same over the entries of the "
                                             "planted low-rank part (synth.torch_problem), the only part a factor model can predict",
                                     "planted_part": planted_err},
        }
        if one_device or transport:
            out["INVALID_AS_A_MEASUREMENT"] = ("test run: MALS_BENCH_ONE_DEVICE=%s (all ranks on device 0), MALS_BENCH_TRANSPORT=%s"
                                               % (os.environ.get("MALS_BENCH_ONE_DEVICE", "0"), transport or ""))
        big = args.workload == "c5"    # no second problem / second arithmetic next to 200 GB of resident data
        if world == 1 and not rank_shape and not force and args.planted > 0 and prob.get("planted") and not args.no_unplanted:
            out["roofline_unplanted"] = unplanted_leg(torch, pkg, sharded, synth, n_users, n_items, nnz_req, k, args, gmode, smode, local_rank, device)
            out["roofline_unplanted"]["slower_than_planted_by"] = out["roofline_unplanted"]["ms_per_step"] / ms_per_step - 1.0
            if dom == "rows":
                un, ums = out["roofline_unplanted"].pop("_all_rows_launches")
                tally["rows"][0] += un
                tally["rows"][1] += ums
            out["roofline_unplanted"].pop("_all_rows_launches", None)
        # what a rocprofv3 --kernel-trace --stats run of THIS command averages for the dominant kernel: every launch of the
        # process (warm-up, timed, the untimed per-half pass, and the un-planted leg -- same kernel, same grid)
        out["roofline"]["all_launches_in_process"] = {"launches": tally[dom][0], "avg_ms": tally[dom][1] / max(tally[dom][0], 1),
                                                      "note": "compare with profiles/*_kernel_stats.txt; avg_launch_ms above is the timed region of the headline workload only"}
        # the arithmetic closest to what "dtype": "f32" promises: the same workload with fp32 products in the per-row Gramian
        # (v_mfma_f32_16x16x4_f32), so that the cost of the split-f16 choice is in the record
        if world == 1 and not rank_shape and not force and split and args.gramian_mode == "auto" and not args.no_fp32_leg and not big:
            leg = unplanted_leg(torch, pkg, sharded, synth, n_users, n_items, nnz_req, k, args, 1, smode, local_rank, device, prob=prob,
                                label="the headline workload with --gramian-mode fp32 (per-row Gramian on v_mfma_f32_16x16x4_f32)")
            leg.pop("_all_rows_launches", None)
            leg["kernel"] = "rows kernel (als_persistent_kernel, MODE 0, fp32 gather)"
            leg["slower_than_split_f16_by"] = leg["ms_per_step"] / ms_per_step - 1.0
            out["roofline_fp32"] = leg
            # ... and the arithmetic in between: THREE f16 terms per operand (every fp32 operand exactly, six products per tile on
            # v_mfma_f32_16x16x32_f16) -- fp32-operand accuracy on the f16 pipe; features 49..64 only
            if 49 <= k <= 64:
                leg = unplanted_leg(torch, pkg, sharded, synth, n_users, n_items, nnz_req, k, args, 3, smode, local_rank, device, prob=prob,
                                    label="the headline workload with --gramian-mode split3_f16 (three f16 terms per operand, 24+ bits, six MFMAs per tile)")
                leg.pop("_all_rows_launches", None)
                leg["kernel"] = "rows kernel (als_persistent_kernel_h<4,0,.,3>, MODE 0)"
                leg["slower_than_split_f16_by"] = leg["ms_per_step"] / ms_per_step - 1.0
                out["roofline_split24"] = leg
        if world == 1 and not args.no_cpu_baseline and not rank_shape and not big:
            X = als.factors(pkg.SIDE_X)
            Y = als.factors(pkg.SIDE_Y)
            out["cpu_baseline"] = cpu_baseline(torch, pkg, prob, X, Y, n_users, n_items, k)
            jvm = jvm_baseline(torch, prob, n_users, n_items, k)
            if jvm is not None:
                out["cpu_baseline_reference_jvm"] = jvm
    # The JSON line must be the LAST thing on the job's stdout.  RCCL prints a banner through C stdio, which sits in
    # every rank's buffer until that process exits -- i.e. after rank 0 has printed.  So: every rank flushes its C
    # stdio, all ranks meet, rank 0 prints, all ranks meet again and leave without running exit handlers that could
    # still write.
    def flush_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    flush_c_stdio()
    if world > 1 or force:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force:
        dist.barrier()
        flush_c_stdio()
        os._exit(0)


if __name__ == "__main__":
    main()
