#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X 3D-SIFT extraction path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full extraction (Gaussian pyramid + DoG + extrema + keypoints +
SIFT-rank descriptors, all octaves, records back on the host) of ONE 512^3
float32 blob-field volume that is already resident in HBM.

N = 1: the volume on one GPU (sift3d_extract).  `value` = .key records per second.

N > 1 (round 5): `value`, `ms_per_step` and `scaling` ("strong") are the Z-SLAB
SPLIT OF THAT ONE VOLUME over the N GPUs -- one process per GPU, halo exchange
over RCCL (torch.distributed), DESIGN.md section 6 -- with
`same_bytes_as_single_gpu` beside it: what BASELINE.json's "1/2/4/8-GPU Z-slab
scaling" asks for.  It runs as a child job under a time limit after the ranks'
own measurement; if it fails, the line is still printed (`value` null, the child's
exit code and stderr tail under `zslab`) and bench.py exits with code 5 -- a
replica figure is never printed under the north-star label.  Beside it:
  volumes_value / volumes_ms_per_step   every rank extracting its OWN 512^3 volume
                (independent volumes, no collective: weak scaling; rounds 1-4's
                headline), measured first, by all ranks, barrier to barrier;
  zslab_c       the same Z-slab split driven from C in ONE process over all N
                devices (sift3d_zslab_set_volume + sift3d_zslab_extract_resident:
                what `featExtract -d0,1,..` ships), once with peer copies and once
                with RCCL as the transport.
`--config c4` / `--config c5` run BASELINE's configs C4 (1024 x 1024 x 512,
N = 4) and C5 (2048 x 2048 x 1024, NRRIEF, N = 8) through the same code; for
those only rank 0 extracts the whole volume (the single-GPU records the merged
ones are compared with), there is no replica leg.

Rank 0 prints ONE JSON line: metric = keypoints/s (.key records per second,
whole job), plus
  roofline     the dominant pyramid kernel (the fused x+y+z+DoG blur, one launch per
               level): algorithmic = compulsory bytes (12 B/voxel,
               8 where only the level or only the DoG is kept) / launch time over its
               512^3 launches, measured here with HIP events on the stream the kernels
               run on; per tap count in `per_instantiation`; `ceiling_*` = what a
               zero-arithmetic march of the same tiles and write streams sustains in
               this process on this box (tools/roof_lib.hip)
  pyramid      Gauss-pyramid + DoG GB/s over all blur launches of a step
  cpu_baseline the CPU restatement (oracle/, single thread like the reference's
               extractor) timed on this box on the 512^3 metric volume, N = 1 only;
               cpu_baseline_ncores: its OpenMP build on all host cores of the box
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_TABLE = "r06_pmc_traffic.json"   # rocprofv3 --pmc passes of this build (tools/make_profiles.sh)
BLUR_SOURCE = os.path.join("3d_sift_cuda_amd", "csrc", "kernels_blur_fused.hip")   # what the PMC table was measured on


# BASELINE.json configs that bench.py can run at N > 1 besides the metric volume: (nx, ny, nz, descriptor mode)
CONFIGS = {"c4": (1024, 1024, 512, 0), "c5": (2048, 2048, 1024, 3)}


def resolve_volume(args):
    """(nx, ny, nz, desc, label) of the run: --config, else --dims, else the cube of edge --size."""
    if args.config:
        nx, ny, nz, desc = CONFIGS[args.config]
    elif args.dims:
        nx, ny, nz = (int(v) for v in args.dims.split(","))
        desc = None
    else:
        nx = ny = nz = args.size
        desc = None
    if args.desc is not None:
        desc = args.desc
    if desc is None:
        desc = 0
    label = "%d^3" % nx if nx == ny == nz else "%d x %d x %d" % (nx, ny, nz)
    return nx, ny, nz, desc, label


def blur_source_hash():
    import hashlib
    with open(os.path.join(ROOT, BLUR_SOURCE), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def kernel_name(stage, ntaps, dog, vec=4):
    r = ntaps // 2
    if stage == "blur_fused":
        return "blur_fused_ring_kernel<%d, rows per thread, has level, %s, prefetch planes, tile x, tile y, has half-size volume, staggered halves>" % (r, "true" if dog else "false")
    if stage == "blur_x":
        return "blur_x_kernel<%d,%d>" % (r, vec)
    return "blur_col_kernel<%d,%d,%s>" % (r, vec, "true" if dog else "false")


def records_match(gpu, cpu):
    """The GPU step's records against the CPU restatement's on the same volume (both in memory in the cpu_baseline leg):
    integer fields and descriptors exact, float fields by largest absolute difference, and whether every byte agrees."""
    res = {"gpu_records": int(len(gpu)), "cpu_records": int(len(cpu)), "same_count": bool(len(gpu) == len(cpu))}
    if len(gpu) != len(cpu):
        res["match"] = False
        return res
    res["info_equal"] = bool((gpu["info"] == cpu["info"]).all())
    res["desc_equal"] = bool((gpu["desc"] == cpu["desc"]).all())
    res["max_abs_diff"] = {f: float(np.abs(gpu[f].astype(np.float64) - cpu[f].astype(np.float64)).max()) if len(gpu) else 0.0
                           for f in ("x", "y", "z", "scale", "ori", "eigs")}
    res["bit_identical"] = bool(gpu.tobytes() == cpu.tobytes())
    res["tolerance"] = 1e-4
    res["match"] = bool(res["info_equal"] and res["desc_equal"] and all(v <= 1e-4 for v in res["max_abs_diff"].values()))
    return res


class Watchdog:
    """Per-phase time limit for a rank of a multi-process run.  `phase(name, limit_s)` names what the rank is about to do and
    how long it may take; a daemon thread ends the PROCESS (exit code 3) with the phase name and every thread's Python stack
    on stderr when the limit passes.  torch.distributed.run then ends the other ranks and returns non-zero, so a stall in any
    phase of the one multi-rank path becomes a named failure within its limit instead of a silent hang (round-3 review,
    weak 7: one of five two-rank rehearsals hung in the main measurement, cause unknown)."""

    def __init__(self, rank):
        import threading
        self.rank, self.name, self.deadline, self.t0 = rank, None, None, time.monotonic()
        self.lock = threading.Lock()
        self.history = []
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def phase(self, name, limit_s):
        with self.lock:
            now = time.monotonic()
            if self.name is not None:
                self.history.append((self.name, round(now - self.t0, 3)))
            self.name, self.t0 = name, now
            self.deadline = None if not limit_s else now + limit_s

    def done(self):
        self.phase(None, 0)

    def _watch(self):
        import faulthandler
        while True:
            time.sleep(0.25)
            with self.lock:
                name, deadline, t0 = self.name, self.deadline, self.t0
            if name is not None and deadline is not None and time.monotonic() > deadline:
                sys.stderr.write("bench.py: rank %d: phase '%s' exceeded its limit (%.0f s); phases so far: %s\n"
                                 % (self.rank, name, time.monotonic() - t0, self.history))
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                os._exit(3)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks HERE, as a child job under
    torch.distributed.run, relay its output and leave with its exit code.  Done before torch is imported and before anything
    touches the GPU (this process never initialises it: it only waits for the child).  Without this the run would read
    WORLD_SIZE = 1 and measure ONE GPU under an N-GPU label."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env["SIFT3D_BENCH_SELF_LAUNCHED"] = "1"
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd[1:9])))
    sys.stderr.flush()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for l in p.stdout:
        if l.startswith("{") and '"metric"' in l:
            line = l.rstrip("\n")      # the ranks' ONE line: printed last, after the child has ended
        else:
            sys.stderr.write(l)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the launched job ended without a result line\n")
        rc = 4
    return rc


def zslab_measure(args, pkg, torch, dist, rank, world, local_rank, expect=None, phase=lambda *a: None):
    """ONE volume cut into Z-slabs, one slab per rank; step = whole extraction with all its records on rank 0 at the end (stored there
    by every rank's descriptor kernel through one shared list, or -- `--zslab-gather` -- gathered through the backend and merged).
    Returns the result object on rank 0 (None elsewhere).  expect: the single-GPU records of the same volume, if the
    caller has them (rank 0), to state whether the merged records are the same bytes."""
    if args.zslab_inject == "torch-leg":   # (a test: what a rank dying in RCCL set-up looks like to the parent)
        sys.stderr.write("bench.py: rank %d: injected failure of the per-process Z-slab job\n" % rank)
        sys.exit(9)
    phase("zslab: plan, slab context, upload")
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    nx, ny, nz, desc, label = resolve_volume(args)
    ndev = torch.cuda.device_count()
    dev = local_rank % max(1, ndev)
    plan = zs.SlabPlan(nx, ny, nz, world)
    i0, i1 = plan.input_range(rank)
    ctx = pkg.Context(nx, ny, zs.slab_context_slices(plan, rank), device=dev, slab=True)   # no level buffers of its own
    be = zs.HipBackend(pkg, ctx, torch)
    # rank 0: the octaves below the sharded ones (gathered there) on a second context and stream, queued by a second host thread, so that
    # their chain of small launches runs beside rank 0's per-keypoint stage instead of in front of it (--zslab-coarse-inline: as before)
    cdims = None if args.zslab_coarse_inline else zs.coarse_octave_dims(plan)
    cctx = pkg.Context(cdims[0], cdims[1], cdims[2], device=dev, slab=True) if (rank == 0 and cdims) else None
    cbe = zs.HipBackend(pkg, cctx, torch) if cctx is not None else None
    # the deferred patch-halo batch on a communicator of its own, so that it cannot queue in front of a level's halo
    # (--zslab-one-group: no second communicator; the deferred batch then queues behind the per-level halos on the first -- the
    # fallback to try in the same lease if a first run on real links stalls with two communicators per device)
    dgroup = None if args.zslab_one_group else dist.new_group(ranks=list(range(world)), backend=dist.get_backend())
    # one collective on each group before the first halo: with RCCL that creates both communicators here, on every rank at
    # the same point, instead of inside the first point-to-point batch (where a rank talks to one or two neighbours only
    # and the ranks would be setting up connections in different orders)
    phase("zslab: first collective on both process groups")
    warm = torch.zeros(1, device="cpu" if dist.get_backend() == "gloo" else "cuda:%d" % dev)
    dist.all_reduce(warm)
    if dgroup is not None:
        dist.all_reduce(warm, group=dgroup)
    # the rank's input slices live in HBM before timing starts, as the volume of the per-GPU run does
    # (only its own slices are generated: the same bits as those planes of the whole volume, without the 16 GB of config C5)
    slab = torch.from_numpy(pkg.synth_blobs_slices(nx, ny, nz, i0, i1, seed=12345)).to("cuda:%d" % dev)
    torch.cuda.synchronize(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    # Where the records end up (round 5).  `shared`: ONE list in shared memory that every rank's process maps and registers with its
    # device; a rank's descriptor kernel stores its records at their places in the single-GPU order (a 193-word all_gather of the
    # ranks' records per group is all that is exchanged), so there is no gather and no merge behind the last kernel.  Without it
    # (--zslab-gather, or if the list cannot be set up): every rank's records travel to rank 0 through the collective backend -- with
    # RCCL an upload of what was just downloaded, the gather, and a download of all of it on rank 0 -- and are merged on the host.
    shared = [None]
    why_gathered = ["--zslab-gather" if args.zslab_gather else None]
    cdev = "cuda:%d" % dev

    def step():
        with be.stream_scope():
            ex = zs.ZSlabExtractor(be, plan, rank, dist, deferred_group=dgroup, coarse_backend=cbe)
            ex.run(slab, i0)
            if shared[0] is not None:
                n, own = ex.describe_into(shared[0], desc_mode=desc, device=cdev)
                if n is not None:
                    return ex, (shared[0].view(n) if rank == 0 else None), "placed"
                recs, grp = own                                  # the list is too small for this volume's records: the old way
            else:
                recs, grp = ex.describe(desc_mode=desc, copy=False)   # views of the pinned download buffers
            merged = zs.gather_records(dist, rank, world, recs, grp, cdev, dtype=pkg.FEATURE_DTYPE)
        return ex, merged, "gathered"

    phase("zslab: warm-up steps (first halo exchange)")
    ex, merged, how = step()
    if not args.zslab_gather:
        phase("zslab: the shared record list")
        total = [len(merged) if (rank == 0 and merged is not None) else 0]
        dist.broadcast_object_list(total, src=0)
        ok = 1
        real_register = pkg.host_register
        if args.zslab_inject == "shared-list" and rank == 1:
            def refuse(address, nbytes):
                raise pkg.Sift3DError("injected: hipHostRegister of the foreign segment refused on this rank")
            pkg.host_register = refuse
        try:
            shared[0] = zs.SharedRecordList(pkg, dist, rank, total[0] + total[0] // 8 + 4096, pkg.FEATURE_DTYPE)
        except Exception as e:   # (a rank that cannot map or register the list: every rank goes back to the gather)
            sys.stderr.write("bench.py: rank %d: no shared record list (%s): gathering the records instead\n" % (rank, e))
            ok = 0
            why_gathered[0] = "the shared record list could not be set up (%s)" % (str(e)[:200],)
        finally:
            pkg.host_register = real_register
        flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if shared[0] is not None:
                shared[0].close()
            shared[0] = None
    phase("zslab: warm-up steps")
    for _ in range(max(1, args.warmup)):
        step()
    phase("zslab: barrier before the timed steps")
    barrier()
    phase("zslab: timed steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ex, merged, how = step()
    phase("zslab: barrier after the timed steps")
    barrier()
    elapsed = time.perf_counter() - t0
    phase("zslab: max over ranks, result")
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda:%d" % dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms_per_step = 1e3 * float(el.item()) / args.steps
    # every rank's exchange statistics of the last step, on rank 0 (a few hundred bytes per rank, after the timed region)
    mine = {"rank": rank, "halo_bytes": ex.stats["exchange_bytes"], "halo_bytes_critical": ex.stats["exchange_bytes"] - ex.stats["deferred_bytes"],
            "halo_bytes_hidden": ex.stats["hidden_bytes"], "halo_bytes_deferred": ex.stats["deferred_bytes"],
            "per_octave": {str(o): {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()} for o, d in sorted(ex.stats["per_octave"].items())}}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    res = None
    if rank == 0:
        nrec = 0 if merged is None else len(merged)
        res = {"value": round(nrec / (ms_per_step * 1e-3), 1), "unit": "keypoints/s", "ms_per_step": round(ms_per_step, 3),
               "scaling": "strong", "records": nrec,
               "workload": "ONE %s float32 blob-field volume cut into %d Z-slabs, full featExtract path (%s descriptor), all octaves"
                           % (label, world, ["SIFT-rank", "BRIEF", "RRIEF", "NRRIEF"][desc]),
               "sharded_octaves": plan.n_sharded, "slab_bounds": plan.bounds,
               "parallelism": "zslab%d: halo exchange with torch.distributed (%s, %s), coarse octaves on rank 0 (%s)"
                              % (world, dist.get_backend(), "one communicator" if dgroup is None else "two communicators: per-level halos / deferred patch halos",
                                 "a second context and host thread beside rank 0's per-keypoint stage" if cbe is not None else "in front of its per-keypoint stage"),
               "records_to_rank0": ("placed: every rank's descriptor kernel stores its records at their places of the single-GPU order in ONE "
                                    "shared, device-registered list (193 words per rank exchanged; no gather, no merge)") if how == "placed"
                                   else "gathered: every rank's records through the collective backend to rank 0, merged on the host",
               "halo_exchanges_per_step": ex.stats["exchanges"], "halo_bytes_per_rank_per_step": ex.stats["exchange_bytes"],
               "halo_bytes_critical_per_rank_per_step": ex.stats["exchange_bytes"] - ex.stats["deferred_bytes"],
               "halo_bytes_deferred_per_rank_per_step": ex.stats["deferred_bytes"],
               "halo_bytes_hidden_per_rank_per_step": ex.stats["hidden_bytes"],
               "per_rank": per_rank,
               "per_rank_note": "every rank's halo bytes of one step (critical = waited for by the next launch; hidden = of those, issued bands-first "
                                "and moving while the rank filters its interior; deferred = patch halos on the second communicator) and, per sharded "
                                "octave, the bytes, batches and the host time the rank spent completing them (gloo: the transfer as the rank sees it; "
                                "nccl: a completed wait only orders streams, so this is queueing cost and the transfer shows in the step)",
               "records_gathered_because": why_gathered[0] if how != "placed" else None,
               "exchange_schedule": "per stored level (L1..L4: the 17-tap L5 is only evaluated around candidates, from L4): the 8-slice blur halo "
                                    "(what the next blur needs; 9 slices of L4), issued BANDS FIRST -- a rank filters its two boundary "
                                    "bands, hands them to the exchange and filters its interior while they travel (`hidden` bytes); per "
                                    "octave: two deferred batches on a communicator of their own, issued when L3 is complete -- the 8 slices of "
                                    "L3 the subsample reads (waited for at the octave's end) and the 11 + 15 + 12 slices of L1..L3 only "
                                    "patches reach (waited for at the end of the pyramid) (rank 0's counts; interior ranks exchange on "
                                    "both sides)"}
        import hashlib
        res["records_sha256"] = hashlib.sha256(merged.tobytes()).hexdigest() if merged is not None else None
        if expect is not None:
            res["same_bytes_as_single_gpu"] = bool(merged is not None and len(merged) == len(expect)
                                                   and (merged.view(np.uint8) == expect.view(np.uint8)).all())
    if merged is not None:
        merged = None   # (a view of the shared list: dropped before the list is)
    if shared[0] is not None:
        shared[0].close()
    if cctx is not None:
        cctx.close()
    ctx.close()
    return res


def zslab_main(args, pkg, torch, dist, rank, world, local_rank, phase=lambda *a: None):
    res = zslab_measure(args, pkg, torch, dist, rank, world, local_rank, phase=phase)
    if rank == 0:
        print(json.dumps({
            "metric": "keypoints/s (.key records per second)", "value": res["value"],
            "unit": "keypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {k: res[k] for k in ("workload", "records", "sharded_octaves", "slab_bounds", "parallelism",
                                           "halo_exchanges_per_step", "halo_bytes_per_rank_per_step",
                                           "halo_bytes_critical_per_rank_per_step", "halo_bytes_deferred_per_rank_per_step",
                                           "halo_bytes_hidden_per_rank_per_step", "exchange_schedule", "records_to_rank0", "records_sha256",
                                           "per_rank", "per_rank_note", "records_gathered_because")}}))
    phase("zslab: leaving the process group")
    dist.barrier()
    dist.destroy_process_group()
    phase(None, 0)


def zslab_child(args, world, expect, limit_s):
    """N > 1, default mode, rank 0 only, AFTER the per-GPU-volume measurement is complete and its process group is gone:
    run the Z-slab split of ONE volume over the same GPUs as a fresh child job (`bench.py --mode zslab` under
    torch.distributed.run, its own rendezvous port, its own process group on the host) and return what goes into the line
    as `zslab`.  It is the only place the slab exchange meets RCCL on real links -- the development box has one GPU -- so a
    crash or a stall there is a result to report (status, exit code, the tail of its stderr), not something that may take
    the headline measurement with it: the child is killed as a process group at the limit, the parent job always ends
    normally."""
    import hashlib
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(world), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--dims", "%d,%d,%d" % resolve_volume(args)[:3], "--desc", str(resolve_volume(args)[3]), "--mode", "zslab",
           "--phase-limit", str(max(1, min(args.phase_limit, limit_s)) if args.phase_limit > 0 else 0)] + (["--zslab-one-group"] if args.zslab_one_group else []) \
        + (["--zslab-gather"] if args.zslab_gather else []) + (["--zslab-coarse-inline"] if args.zslab_coarse_inline else []) \
        + (["--zslab-inject", args.zslab_inject] if args.zslab_inject else [])
    # End exactly the job started here (run_child): torch.distributed.run puts every rank in a session of its own, so the
    # launcher's process group does not contain them -- their PIDs are noted first, the launcher is asked to stop (it
    # terminates its ranks on SIGTERM), then whatever of it is still there is killed.
    rc, so, se = run_child(cmd, child_env(), limit_s)
    if rc is None:
        return {"status": "no result within %d s (a rank failed or the exchange stalled); the child job was killed" % limit_s,
                "exit_code": None, "stderr_tail": se[-600:]}
    line = None
    for l in so.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
    if rc != 0 or line is None:
        return {"status": "the child job failed", "exit_code": rc, "stderr_tail": se[-600:]}
    d = json.loads(line)
    res = {"status": "ok", "exit_code": 0, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "scaling": "strong"}
    res.update(d["config"])
    if expect is not None:
        res["same_bytes_as_single_gpu"] = bool(res.get("records_sha256") == hashlib.sha256(expect.tobytes()).hexdigest())
    return res


def zslab_c_main(args, pkg):
    """`--mode zslab_c` (ONE process, no launcher): the Z-slab split of the volume driven from C over `--gpus` devices --
    sift3d_zslab_create, sift3d_zslab_set_volume once, then sift3d_zslab_extract_resident per step: the driver
    `featExtract -d0,1,..` ships, in its resident form (round-4 review item 1b).  Once with peer copies, once with RCCL
    (ncclSend / ncclRecv in groups, two communicator sets).  One JSON line per transport as soon as it is measured, so that a
    stall in the second leaves the first on record; the parent embeds them as `zslab_c`.  A step is bracketed by the call
    itself: it returns when every device's streams have drained and the records are merged on the host."""
    import hashlib
    nx, ny, nz, desc, label = resolve_volume(args)
    ndev = pkg.device_count()
    if ndev < 1:
        raise SystemExit("bench.py --mode zslab_c: no HIP device")
    N = args.gpus
    devices = list(range(N)) if ndev >= N else [i % ndev for i in range(N)]   # fewer devices than ranks: the one-GPU rehearsal
    rehearsal_lib = os.environ.get("SIFT3D_BENCH_RCCL_LIBRARY")               # tests/rccl_shim on a one-GPU box
    vol = pkg.synth_blobs(nx, ny, nz, seed=12345)
    with pkg.ZSlab(nx, ny, nz, devices) as h:
        h.set_volume(vol)
        del vol
        for name, tr in (("peer_copy", pkg.TRANSPORT_PEER_COPY), ("rccl", pkg.TRANSPORT_RCCL)):
            res = {"transport_asked": name, "devices": devices}
            try:
                if tr == pkg.TRANSPORT_RCCL and rehearsal_lib:
                    pkg.zslab_set_transport_library(rehearsal_lib)
                    h.set_tuning(pkg.ZSLAB_DUPLICATE_RANKS, 1)
                    res["rccl_library"] = rehearsal_lib
                h.set_tuning(pkg.ZSLAB_TRANSPORT, tr)
                h.set_tuning(pkg.ZSLAB_SERIAL_CHANNELS, 1 if args.zslab_one_group else 0)
                for _ in range(max(1, args.warmup)):
                    recs, st = h.extract_resident(desc_mode=desc, copy=False)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    recs, st = h.extract_resident(desc_mode=desc, copy=False)
                ms = 1e3 * (time.perf_counter() - t0) / args.steps
                res.update({"status": "ok", "value": round(len(recs) / (ms * 1e-3), 1), "unit": "keypoints/s", "ms_per_step": round(ms, 3),
                            "scaling": "strong", "records": int(len(recs)), "records_sha256": hashlib.sha256(recs.tobytes()).hexdigest(),
                            "n_ranks": st["n_ranks"], "sharded_octaves": st["sharded_octaves"],
                            "transport": "rccl" if st["transport"] == pkg.TRANSPORT_RCCL else "peer_copy",
                            "transport_fell_back": bool(st["transport_fell_back"]), "rccl_version": st["rccl_version"], "comm_sets": st["comm_sets"],
                            "halo_bytes_critical": st["halo_bytes_critical"], "halo_bytes_hidden": st["halo_bytes_hidden"],
                            "halo_bytes_deferred": st["halo_bytes_deferred"], "halo_bytes_subsample": st["halo_bytes_subsample"],
                            "gather_bytes": st["gather_bytes"], "merge_ms": round(st["merge_ms"], 3),
                            "enqueue_ms": round(st["enqueue_ms"], 3),   # host time until the last rank's thread had queued its pyramid and count request
                            "resident_volume": bool(st["resident_volume"]),
                            "workload": "ONE %s volume, %d Z-slabs, one process over devices %s (sift3d_zslab_extract_resident)" % (label, N, devices)})
            except pkg.Sift3DError as e:
                res.update({"status": "failed", "error": str(e)[-400:]})
            print(json.dumps({"zslab_c": res}), flush=True)


def run_child(cmd, env, limit_s):
    """Start cmd in a session of its own, wait at most limit_s, end exactly what was started.  Returns (returncode or None,
    stdout, stderr)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        so, se = p.communicate(timeout=limit_s)
        return p.returncode, so, se
    except subprocess.TimeoutExpired:
        try:
            import psutil
            try:
                kids = psutil.Process(p.pid).children(recursive=True)
            except psutil.NoSuchProcess:
                kids = []
        except ImportError:
            psutil, kids = None, []
        p.send_signal(signal.SIGTERM)
        try:
            so, se = p.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        for k in kids:
            try:
                if k.is_running():
                    k.kill()
            except Exception:
                pass
        return None, so, se


def child_env():
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
            "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC"))}


def zslab_c_child(args, world, expect, limit_s):
    """rank 0, after the process group is gone: `bench.py --mode zslab_c --gpus N` as ONE child process (no launcher)."""
    import hashlib
    nx, ny, nz, desc, _ = resolve_volume(args)
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "zslab_c", "--gpus", str(world), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--dims", "%d,%d,%d" % (nx, ny, nz), "--desc", str(desc)] + (["--zslab-one-group"] if args.zslab_one_group else [])
    rc, so, se = run_child(cmd, child_env(), limit_s)
    out = {}
    for l in so.splitlines():
        if l.startswith("{") and '"zslab_c"' in l:
            r = json.loads(l)["zslab_c"]
            if expect is not None and r.get("records_sha256"):
                r["same_bytes_as_single_gpu"] = bool(r["records_sha256"] == hashlib.sha256(expect.tobytes()).hexdigest())
            out[r.pop("transport_asked")] = r
    out["status"] = "ok" if rc == 0 else ("no result within %d s; the child was killed" % limit_s if rc is None else "the child failed")
    out["exit_code"] = rc
    if rc != 0:
        out["stderr_tail"] = se[-600:]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512, help="edge of the cubic volume (512 = the BASELINE metric)")
    ap.add_argument("--dims", default=None, metavar="NX,NY,NZ", help="a volume that is not a cube (overrides --size)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="a BASELINE.json configuration instead of the metric volume: c4 = 1024 x 1024 x 512 (start it with --gpus 4), "
                         "c5 = 2048 x 2048 x 1024 with the NRRIEF descriptor (--gpus 8); N > 1 only")
    ap.add_argument("--cpu-sample", type=int, default=512,
                    help="edge of the CPU-baseline volume (0 = skip); 512 = the metric volume itself, about 25 s on one core")
    ap.add_argument("--desc", type=int, default=None, help="0 SIFT-rank (default), 1 BRIEF, 2 RRIEF, 3 NRRIEF")
    ap.add_argument("--mode", default="auto", choices=["auto", "volumes", "zslab", "zslab_c"],
                    help="N > 1: 'auto' (default) = the ranks' own volumes first (weak scaling, `volumes_value`), then the Z-slab "
                         "split of ONE volume as a child job = the line's `value` (strong scaling), then the C driver's resident "
                         "form (`zslab_c`); 'volumes' = only the first; 'zslab' = only the Z-slab run (under a launcher); "
                         "'zslab_c' = only the one-process C driver (no launcher)")
    ap.add_argument("--zslab-limit", type=int, default=180,
                    help="N > 1: seconds each of the two Z-slab child jobs may take")
    ap.add_argument("--zslab-one-group", action="store_true",
                    help="N > 1: the Z-slab run's deferred patch-halo batch on the SAME communicator as the per-level halos (default: a "
                         "second one); for the C driver the same switch is SIFT3D_ZSLAB_SERIAL_CHANNELS")
    ap.add_argument("--zslab-gather", action="store_true",
                    help="N > 1: the Z-slab run's records gathered on rank 0 through the collective backend and merged on the host (rounds 1 - 4) "
                         "instead of stored by every rank's kernel in one shared list")
    ap.add_argument("--zslab-coarse-inline", action="store_true",
                    help="N > 1: the Z-slab run's octaves below the sharded ones on rank 0's own context and thread, in front of its per-keypoint "
                         "stage (rounds 1 - 4), instead of on a second context queued by a second host thread")
    ap.add_argument("--phase-limit", type=int, default=300,
                    help="N > 1: seconds any one phase of a rank (set-up, warm-up, the timed steps, a reduction) may take before "
                         "the rank ends the job with exit code 3 and the phase name (0 = no limit)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="sift3d_set_tuning on the context before the run, e.g. FUSED_TILE=2 (A/B measurements; no knob changes a "
                         "result; the line records what was set)")
    ap.add_argument("--zslab-inject", default=None, choices=["shared-list", "torch-leg"],
                    help="tests of the N > 1 fall-backs: 'shared-list' -- rank 1 fails to register the shared record list, every rank "
                         "must go back to gathering the records and the line must say why; 'torch-leg' -- the per-process Z-slab job "
                         "exits with code 9 before its first collective, the line's value must then be the one-process C driver's")
    ap.add_argument("--launch-check", default=None, choices=["ok", "fail"],
                    help="only bring the ranks up (process group over gloo, no GPU call), print a line with the world size "
                         "seen and leave -- 'fail': rank 1 exits with code 7 instead (tests of the self-launch path)")
    args = ap.parse_args()

    if args.mode == "zslab_c":   # one process over all devices: no launcher, no torch
        pkg = importlib.import_module("3d_sift_cuda_amd")
        return zslab_c_main(args, pkg)
    if args.config and args.gpus == 1:
        raise SystemExit("bench.py: --config %s is a multi-GPU configuration (use --gpus N); on one GPU use --dims" % args.config)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))   # before torch is imported: this process never touches the GPU

    if os.environ.get("BENCH_DUMP_STACKS_AFTER"):   # diagnosis of a hang: every thread's Python stack to stderr after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_DUMP_STACKS_AFTER"]), repeat=False, exit=False)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(rank) if (world > 1 and args.phase_limit > 0) else None

    def phase(name, limit=None):
        if wd is not None:
            wd.phase(name, args.phase_limit if limit is None else limit)

    phase("import torch + process group")
    import torch
    dist = None
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        seen = dist.get_world_size()
        if args.launch_check == "fail" and rank == 1:
            sys.exit(7)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "launch check", "value": 0, "n_gpus": seen, "launch_check": True,
                              "launched_by": "self" if os.environ.get("SIFT3D_BENCH_SELF_LAUNCHED") else "external launcher"}), flush=True)
        dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("SIFT3D_DIST_BACKEND", "nccl")   # "gloo" only for the single-GPU rehearsal of zslab mode
        if backend == "nccl" and world > ndev:
            raise SystemExit("bench.py: %d ranks but %d GPU(s) visible: RCCL needs one device per rank "
                             "(SIFT3D_DIST_BACKEND=gloo rehearses several ranks on one GPU)" % (world, ndev))
        local_rank = local_rank % max(1, ndev)   # (ranks share a GPU only in the gloo rehearsal)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        world = dist.get_world_size()   # what the process group really holds: this is what the line reports as n_gpus
    if args.gpus != world:
        # never a silent relabel: a line that says n_gpus = N must come from N ranks
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s); start it as `python bench.py --gpus %d` "
                         "(it launches its ranks itself) or under torch.distributed.run with --nproc-per-node %d"
                         % (args.gpus, world, args.gpus, args.gpus))

    pkg = importlib.import_module("3d_sift_cuda_amd")
    if not os.path.exists(pkg.LIB_HIP):
        raise SystemExit("libsift3d_hip.so missing: run python __graft_entry__.py (no CPU fallback)")
    nx, ny, nz, desc, vol_label = resolve_volume(args)
    args.desc = desc
    n = nx   # (the roofline bookkeeping below is written for the cubic metric volume; other shapes only fill the generic fields)
    nvox = nx * ny * nz
    if args.mode == "zslab" and world > 1:
        return zslab_main(args, pkg, torch, dist, rank, world, local_rank, phase)
    # The ranks' own volumes (weak scaling).  For the large configurations (--config c4 / c5) only rank 0 extracts the whole
    # volume -- the single-GPU records the Z-slab run's merged records are compared with -- and the others wait at the barrier.
    replicas = args.config is None
    if not replicas and rank != 0:
        phase("volumes: waiting for rank 0's single-GPU run")
        dist.barrier(); dist.barrier()   # rank 0's barriers around its timed steps
        z = torch.zeros(1, dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda:%d" % local_rank)
        dist.all_reduce(z, op=dist.ReduceOp.MAX)   # elapsed
        dist.all_reduce(z, op=dist.ReduceOp.SUM)   # records
        phase("volumes: leaving the process group")
        dist.barrier()
        dist.destroy_process_group()
        if wd is not None:
            wd.done()
        return
    phase("volumes: synthetic volume, context, upload")
    vol = pkg.synth_blobs(nx, ny, nz, seed=12345 + (rank if replicas else 0))
    ctx = pkg.Context(nx, ny, nz, device=local_rank)
    for kv in args.tune:
        k, v = kv.split("=")
        ctx.set_tuning(getattr(pkg, "TUNE_" + k.upper()), int(v))
    dvol = torch.from_numpy(vol).to("cuda:%d" % local_rank)   # the input lives in HBM before timing starts
    torch.cuda.synchronize(local_rank)
    ctx.set_volume_dev(dvol.data_ptr(), nx, ny, nz)
    ctx.sync()

    def barrier():
        torch.cuda.synchronize(local_rank)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(local_rank)

    nrec = 0
    phase("volumes: warm-up steps")
    for _ in range(args.warmup):
        nrec = len(ctx.extract(desc_mode=args.desc, copy=False))
    # Timed region: HIP events only around the dominant kernels (the blur launches on the full-size volume: five per
    # step).  Bracketing all ~170 launches of a step costs about 1 ms of the step; the full per-stage breakdown is
    # taken from extra steps after the timed region.
    ctx.enable_timing(2)
    logs = []
    phase("volumes: barrier before the timed steps")
    barrier()
    phase("volumes: timed steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats = ctx.extract(desc_mode=args.desc, copy=False)   # records land in pinned host memory
        nrec = len(feats)
        logs.append(ctx.launch_log())
    ctx.sync()
    phase("volumes: barrier after the timed steps")
    barrier()
    elapsed = time.perf_counter() - t0
    phase("volumes: breakdown steps")
    tim = ctx.timings()
    gpu_recs = feats.copy() if rank == 0 else None   # the timed region's last step (the pinned view is reused by the next call)
    full_logs, excl_logs = [], []
    if rank == 0:
        # The breakdown steps run the one-list schedule (SIFT3D_TUNE_SPLIT_TAIL 0): on the production schedule the launches that
        # build the coarse octaves queue behind the first part's keypoint kernel, and an event pair around such a launch
        # times that wait (3 ms), not the kernel.  The timed region above ran the production schedule.
        split_was = 1
        for kv in args.tune:
            if kv.split("=")[0].upper() == "SPLIT_TAIL":
                split_was = int(kv.split("=")[1])
        ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, 0)
        ctx.enable_timing(1)
        for _ in range(2):
            ctx.extract(desc_mode=args.desc, copy=False)
            full_logs.append(ctx.launch_log())
        ctx.enable_timing(3)   # the same with the extrema kept on the main stream: every launch timed alone
        for _ in range(2):
            ctx.extract(desc_mode=args.desc, copy=False)
            excl_logs.append(ctx.launch_log())
        ctx.enable_timing(0)
        ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, split_was)

    phase("volumes: max / sum over ranks")
    red_dev = "cpu" if (dist is not None and dist.get_backend() == "gloo") else "cuda:%d" % local_rank   # gloo: the one-GPU rehearsal
    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rc = torch.tensor([float(nrec)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rc, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_records = float(rc.item())
    ms_per_step = 1e3 * elapsed / args.steps
    phase("volumes: result line (rank 0: breakdown, CPU baseline)", 0)

    if rank == 0:
        nfullvox = ((nx + 3) // 4 * 4) * ny * nz   # rows are padded to whole 16-byte vectors inside the pipeline
        log = np.concatenate(logs)            # timed region: the blur launches on the n^3 volume
        full = np.concatenate(full_logs)      # two extra steps with every launch bracketed (per-stage breakdown)
        nfull = len(full_logs)
        stage_names = pkg.STAGES
        # ---- per-kernel grouping: (stage, ntaps) over the breakdown steps ----
        groups = {}
        for r in full:
            st = stage_names[r["stage"]]
            key = (st, int(r["ntaps"]), bool(st in ("blur_z_dog", "blur_fused") and r["alg_bytes"] > 8.5 * r["nvox"]))
            g = groups.setdefault(key, {"ms": 0.0, "bytes": 0.0, "launches": 0})
            g["ms"] += float(r["ms"]); g["bytes"] += float(r["alg_bytes"]); g["launches"] += 1
        blur_groups = {k: v for k, v in groups.items() if k[0].startswith("blur") or k[0] == "octave_tiny"}
        # The dominant kernel of the pyramid is the fused blur (one template, one instantiation per tap count); its
        # roofline figure is taken over its n^3 (octave-0) launches -- 7/8 of the pyramid's bytes, and they run before
        # anything shares the chip with them (the extrema of an octave overlap the blurs of the coarser ones).  The
        # same kernels also run on the coarser octaves; that aggregate is what `rocprofv3 --stats` averages.
        fused_id = stage_names.index("blur_fused") if "blur_fused" in stage_names else -1
        sel = log[(log["stage"] == fused_id) & (log["nvox"] == nfullvox)]
        if len(sel):
            dom_name = ("blur_fused_ring_kernel<R, rows per thread, has level, has DoG, prefetch planes, tile x, tile y, has half-size volume, staggered halves> "
                        "(the %d launches per volume at %d^3: initial blur + the levels stored in full -- L1..L4 by default, the "
                        "17-tap level L5 only exists around the candidates of D3 (extrema_validate_lazy_kernel); two rows per "
                        "thread, two or three planes of prefetch, one workgroup per CU; from 11 taps up the second half of the wavefronts half a step behind the first)" % (len(sel) // args.steps, n))
            if not (nx == ny == nz):
                dom_name = dom_name.replace("%d^3" % n, vol_label)
            dom_all = full[full["stage"] == fused_id]
            per_inst = []
            for taps in sorted(set(int(t) for t in sel["ntaps"])):
                for dogflag in (False, True):
                    q = sel[(sel["ntaps"] == taps) & ((sel["alg_bytes"] > 8.5 * sel["nvox"]) == dogflag)]
                    if len(q):
                        per_inst.append({"taps": taps, "alg_bytes_per_voxel": round(float(q["alg_bytes"][0]) / nvox, 1),
                                         "launches": int(len(q)), "avg_launch_ms": round(float(q["ms"].mean()), 4),
                                         "GBs": round(float(q["alg_bytes"].sum()) / (float(q["ms"].sum()) * 1e-3) / 1e9, 1)})
            accounting = ("fused x+y+z+DoG launches: compulsory bytes only -- read the level once, write what is kept (level and "
                          "DoG: 12 B/voxel; level 3 also writes the next octave's level 0: 12.5; initial blur and L1, whose DoG is not stored: level only, 8 B/voxel); the three-pass "
                          "form of the same work is credited 32 (24) B/voxel")
        else:   # rows that are not whole 16-byte vectors: three-pass kernels; dominant = slowest instantiation at n^3
            big = {}
            for r in log[log["nvox"] == nfullvox]:
                st = stage_names[r["stage"]]
                if st.startswith("blur"):
                    key = (st, int(r["ntaps"]), bool(st == "blur_z_dog" and r["alg_bytes"] > 8.5 * r["nvox"]))
                    big[key] = big.get(key, 0.0) + float(r["ms"])
            dom_key = max(big, key=big.get)
            sel = log[(log["stage"] == stage_names.index(dom_key[0])) & (log["ntaps"] == dom_key[1]) & (log["nvox"] == nfullvox)]
            if dom_key[0] == "blur_z_dog":
                sel = sel[(sel["alg_bytes"] > 8.5 * sel["nvox"]) == dom_key[2]]
            dom_name = kernel_name(dom_key[0], dom_key[1], dom_key[2], 4 if n % 4 == 0 else 1)
            dom_all = full[(full["stage"] == stage_names.index(dom_key[0])) & (full["ntaps"] == dom_key[1])]
            per_inst = None
            accounting = "8 B/voxel per x or y pass, 16 B/voxel for the z pass with fused DoG store"
        big_ms, big_bytes = float(sel["ms"].sum()), float(sel["alg_bytes"].sum())
        achieved = big_bytes / (big_ms * 1e-3) / 1e9
        traffic, traffic_note, traffic_when = None, None, ""
        pmc = os.path.join(ROOT, "profiles", PMC_TABLE)
        if not os.path.exists(pmc):
            traffic_note = "no PMC table profiles/%s" % PMC_TABLE
        elif (nx, ny, nz) != (512, 512, 512) or not per_inst:
            traffic_note = "the PMC table holds 512^3 launches of the fused kernels only"
        else:
            try:   # PMC bytes of every instantiation (keys "blur_fused_ring_kernel<R, ...>"), averaged over the launches
                tab = json.load(open(pmc))
                # the table is only valid for the kernel source it was measured on: a stale table must not be quoted
                if tab.get("_kernel_source_sha256") != blur_source_hash():
                    raise LookupError("profiles/%s was measured on another version of %s (regenerate with tools/make_profiles.sh)"
                                      % (PMC_TABLE, BLUR_SOURCE))
                tot, cnt = 0.0, 0
                for pi in per_inst:
                    R = pi["taps"] // 2
                    with_dog = pi["alg_bytes_per_voxel"] > 9
                    with_sub = pi["alg_bytes_per_voxel"] > 12.2   # the level-3 launch that also writes the next octave's level 0
                    # template arguments: <R, rows per thread, has level, has DoG, prefetch planes, tile x, tile y, has half-size, staggered>
                    def targs(k):
                        a = [v.strip() for v in k[k.index("<") + 1:k.rindex(">")].split(",")]
                        return a + ["false"] * (9 - len(a))
                    mine = [k for k in tab if k.startswith("blur_fused_ring_kernel<%d," % R) and (targs(k)[7] == "true") == with_sub]
                    exact = [k for k in mine if targs(k)[2] == "true" and (targs(k)[3] == "true") == with_dog]
                    twin = [k for k in mine if targs(k)[2] == "true" and targs(k)[3] == "true"]
                    if exact:
                        w = tab[sorted(exact)[0]]["hbm_bytes_per_launch_512"]
                    elif twin:   # a level-only launch measured through its level + DoG twin: 4 B/voxel less written
                        w = tab[sorted(twin)[0]]["hbm_bytes_per_launch_512"] - 4.0 * nvox
                    else:
                        raise LookupError("no PMC entry for %d taps" % pi["taps"])
                    tot += w * pi["launches"]; cnt += pi["launches"]
                traffic = tot / cnt
                m = tab.get("_measured") or {}
                traffic_when = "measured %s UTC on %s (%s) by %s" % (m.get("date_utc", "?"), m.get("host", "?"), m.get("gpu", "?"), m.get("by", "?"))
            except Exception as e:
                traffic, traffic_note = None, "dropped: %s" % (e,)
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": dom_name,
                    "launch": "%s volume (octave 0)" % vol_label, "launches": int(len(sel)),
                    "avg_launch_ms": round(big_ms / max(1, len(sel)), 4),
                    "alg_bytes_per_launch": big_bytes / max(1, len(sel)),
                    "alg_bytes_per_voxel": round(big_bytes / max(1, len(sel)) / nvox, 2),
                    "accounting": accounting,
                    "per_instantiation": per_inst,
                    "all_launches": {"launches": int(len(dom_all)), "avg_launch_ms": round(float(dom_all["ms"].mean()), 4),
                                     "achieved": round(float(dom_all["alg_bytes"].sum()) / (float(dom_all["ms"].sum()) * 1e-3) / 1e9, 1),
                                     "note": "every octave the kernel runs on, from the two breakdown steps after the timed region; compare with the per-kernel averages of rocprofv3 --stats"},
                    "traffic_source": ("profiles/" + PMC_TABLE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled per "
                                       "MI355X_MICROARCH.md), averaged over the launches; a table committed with the source it was measured on "
                                       "(hash-checked), NOT counters of this run: " + traffic_when) if traffic else traffic_note}
        # ---- the ceiling of the kernel's ACCESS SHAPE, measured now, in this process, on this box (round-4 review item 4):
        # the zero-arithmetic march of the same tiles (tools/roof_lib.hip: 64 x 32 and 128 x 16 tiles, two z chunks, one
        # workgroup per CU, one plane read and one or two planes stored per step).  Per instantiation the better of the two
        # tiles for its store streams; the sum of those march times over the step's launches is what a kernel that does
        # nothing but move these tiles would take.  frac_of_ceiling = frac / ceiling_frac.
        roofline["ceiling"] = None
        if per_inst and (nx, ny, nz) == (512, 512, 512):
            try:
                import ctypes
                lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libroof.so"))
                lib.roof_march.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
                torch.cuda.synchronize(local_rank)
                ms4 = (ctypes.c_float * 4)()
                rc_roof = lib.roof_march(512, 10, ms4)
                if rc_roof != 0:
                    raise RuntimeError("roof_march -> %d" % rc_roof)
                m1, m2 = min(ms4[0], ms4[1]), min(ms4[2], ms4[3])   # one / two store streams, the better tile
                c_ms, c_bytes, rows = 0.0, 0.0, []
                for pi in per_inst:
                    bpv = pi["alg_bytes_per_voxel"]
                    march = m1 if bpv < 9 else m2 * (bpv / 12.0)     # 12.5 B/voxel (the half-size volume rides along): the 12-byte march, scaled
                    c_ms += march * pi["launches"]; c_bytes += bpv * nvox * pi["launches"]
                    rows.append({"taps": pi["taps"], "alg_bytes_per_voxel": bpv, "march_ms": round(march, 4), "kernel_ms": pi["avg_launch_ms"],
                                 "kernel_over_march": round(pi["avg_launch_ms"] / march, 3)})
                c_gbs = c_bytes / (c_ms * 1e-3) / 1e9
                roofline["ceiling"] = {
                    "what": "zero-arithmetic march of the same tiles (tools/roof_lib.hip), measured in this process after the timed region: "
                            "what a kernel that only moves the fused blur's tiles sustains on this box (the best of 10 launches and of three "
                            "placements of its buffers per tile shape and store count)",
                    "achieved": round(c_gbs, 1), "unit": "GB/s", "frac": round(c_gbs / HBM_PEAK_GBS, 4),
                    "march_ms": {"64x32_1_store": round(ms4[0], 4), "128x16_1_store": round(ms4[1], 4),
                                 "64x32_2_stores": round(ms4[2], 4), "128x16_2_stores": round(ms4[3], 4)},
                    "per_instantiation": rows}
                roofline["ceiling_frac"] = roofline["ceiling"]["frac"]
                roofline["frac_of_ceiling"] = round(roofline["frac"] / roofline["ceiling"]["frac"], 4)
            except Exception as e:   # a measurement aid: its absence must not cost the line
                roofline["ceiling"] = {"note": "not measured: %r (tools/_build/libroof.so is built by __graft_entry__.build())" % (e,)}
        pyr_ms = sum(v["ms"] for v in blur_groups.values())
        pyr_bytes = sum(v["bytes"] for v in blur_groups.values())
        pyramid = {"alg_GBs": round(pyr_bytes / (pyr_ms * 1e-3) / 1e9, 1), "frac_of_peak": round(pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "ms_per_step": round(pyr_ms / nfull, 3), "alg_bytes_per_step": pyr_bytes / nfull,
                   "accounting": "fused launches (volumes >= 2^22 voxels): 12 B/voxel with DoG, 8 without; three-pass launches "
                                 "(coarse octaves): 8 B/voxel per x or y pass, 16 for the z pass with fused DoG; octaves of at "
                                 "most 4096 voxels: one launch, 40 B/voxel (level 0 in, four levels and five DoGs out)"}
        stages = {}
        for i, s in enumerate(stage_names):
            sel = full[full["stage"] == i]
            if len(sel):
                stages[s] = {"ms_per_step": round(float(sel["ms"].sum()) / nfull, 3), "launches_per_step": len(sel) // nfull}
        stages["_note"] = ("kernel time per stage from two extra steps with every launch bracketed by HIP events (that costs "
                           "about 1 ms per step, so those steps are outside the timed region) on the one-list schedule "
                           "(SIFT3D_TUNE_SPLIT_TAIL 0: one sort, one keypoint and one descriptor launch behind the whole pyramid -- on the "
                           "production schedule the coarse octaves' launches queue behind the first part's keypoint kernel and their "
                           "event pairs would time that wait); the extrema of an octave run "
                           "beside the blurs of the coarser ones, so the stages add up to more than a step and a blur launch's "
                           "event pair also times the extrema kernels that share the chip with it; `exclusive_ms_per_step` is the "
                           "same from two steps in which the extrema stay on the main stream (sift3d_enable_timing mode 3): every "
                           "launch timed alone")
        excl = np.concatenate(excl_logs)
        for i, sname in enumerate(stage_names):
            sel = excl[excl["stage"] == i]
            if len(sel) and sname in stages:
                stages[sname]["exclusive_ms_per_step"] = round(float(sel["ms"].sum()) / len(excl_logs), 3)
        eb = excl[np.isin(excl["stage"], [stage_names.index(q) for q in ("blur_x", "blur_y", "blur_z_dog", "blur_fused", "octave_tiny")])]
        pyramid["exclusive_ms_per_step"] = round(float(eb["ms"].sum()) / len(excl_logs), 3)
        pyramid["exclusive_alg_GBs"] = round(float(eb["alg_bytes"].sum()) / (float(eb["ms"].sum()) * 1e-3) / 1e9, 1)
        pyramid["exclusive_frac_of_peak"] = round(pyramid["exclusive_alg_GBs"] / HBM_PEAK_GBS, 4)
        pyramid["accounting"] += ("; ms_per_step / frac_of_peak: blur launches timed while the finer octave's extrema share the chip "
                                  "(the production schedule); exclusive_*: the same launches timed alone")
        # BASELINE.md section 4 / SURVEY.md section 8d state the pyramid's budget in the bytes of the separable three-pass design
        # the north star names: 24N per blur + 8N per fused DoG = 176N for octave 0, 152N for every later octave (N / 8 each):
        # 197.7N = 26.5 GB at 512^3, "70 % target => <= 4.7 ms for Gaussian + DoG".  The same time priced that way, for
        # comparison with that target only (roofline.* and pyramid.frac_of_peak stay on the compulsory bytes of what is launched):
        n_vox = float(nvox)
        b8d = n_vox * (176.0 + 152.0 / 7.0)
        pyramid["survey_8d_accounting"] = {
            "alg_bytes_per_step": b8d, "target_ms_at_70_percent_of_peak": round(b8d / (0.7 * HBM_PEAK_GBS * 1e9) * 1e3, 3),
            "ms_per_step": pyramid["ms_per_step"], "exclusive_ms_per_step": pyramid["exclusive_ms_per_step"],
            "equivalent_GBs": round(b8d / (pyramid["ms_per_step"] * 1e-3) / 1e9, 1),
            "within_three_pass_time_budget": bool(pyramid["ms_per_step"] <= b8d / (0.7 * HBM_PEAK_GBS * 1e9) * 1e3),
            "note": "a TIME budget, not the roofline target: the three-pass bytes of SURVEY.md 8d (24N per blur, +8N per DoG, all octaves) over the "
                    "time the pyramid's blur launches take here; the fused launches move a third of those bytes, which is why `equivalent_GBs` "
                    "may exceed the HBM peak.  The north star's 0.70 of the HBM roofline is judged on roofline.frac (compulsory bytes of what is "
                    "launched) and is NOT met at 0.61 - 0.62; see roofline.ceiling for what the access shape allows"}
        out = {
            "metric": "keypoints/s (.key records per second; Gauss-pyramid GB/s vs HBM roofline in `pyramid`/`roofline`)",
            "value": round(total_records / (ms_per_step * 1e-3), 1),
            "unit": "keypoints/s",
            "n_gpus": world, "devices_used": min(world, ndev), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s float32 blob-field volume per GPU, full featExtract path (pyramid + DoG + extrema + keypoints + %s descriptor), all octaves"
                                   % (vol_label, ["SIFT-rank", "BRIEF", "RRIEF", "NRRIEF"][args.desc]),
                       "records_per_volume": int(nrec), "octaves": int(tim["n_octaves"]), "extrema": int(tim["n_extrema"]),
                       "keypoints": int(tim["n_keypoints"]),
                       "parallelism": "1 volume per GPU (independent volumes, no collective)" if world > 1 else "single GPU",
                       "tuning": args.tune or "defaults",
                       "descriptor_parity": ("SIFT-rank: pinned against the reference (see DESIGN.md section 2)" if args.desc == 0 else
                                             "BRIEF / RRIEF / NRRIEF exist in the reference only as commented alternatives "
                                             "(MultiScale.cpp:1037-1045): HIP == oracle bit for bit, but the oracle rows are unpinned "
                                             "against the reference")},
            "roofline": roofline, "pyramid": pyramid, "stages": stages,
        }
        if world == 1 and args.cpu_sample > 0:
            # CPU baseline: the C restatement of the reference's CPU path, timed on this box's host
            # cores on the same blob-field volume the GPU step works on (one step of the same workload).  One thread is
            # the baseline proper, because the reference's extractor is single-threaded; the OpenMP build of the same
            # restatement (bit-identical output) is the "fair CPU" line, with the number of threads it used.
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _oracle
            m = args.cpu_sample
            svol = pkg.synth_blobs(m, m, m, seed=12345)
            what = ("the metric volume itself: one step of the benchmark workload" if m == n else
                    "a bounded sample of the workload (the metric volume is %d^3)" % n)

            def timed(orc):
                c0 = time.perf_counter()
                recs, st = orc.extract(svol, desc_mode=args.desc)
                return recs, st, time.perf_counter() - c0
            recs, st, cdt = timed(_oracle.load())
            out["records_match_cpu"] = records_match(gpu_recs, recs) if m == n else None
            out["cpu_baseline"] = {"value": round(len(recs) / cdt, 1), "unit": "keypoints/s", "cores": 1, "kind": "port",
                                   "sample": "oracle o3_extract (C restatement of the reference CPU path, gcc -O2 -ffp-contract=off, 1 thread) on a %d^3 blob-field volume, %s: %d records in %.2f s (blur %.2f s, DoG %.2f s, detect %.2f s, keypoints %.2f s, descriptors %.2f s)"
                                             % (m, what, len(recs), cdt, st.t_blur, st.t_dog, st.t_detect, st.t_features, st.t_desc)}
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            # a one-GPU box of the pool is a 16-CPU share of its host, whatever the affinity mask shows
            ncores = min(ncores, int(os.environ.get("SIFT3D_CPU_THREADS", "16")))
            os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
            try:
                recs2, st2, cdt2 = timed(_oracle.load_omp())
                out["cpu_baseline_ncores"] = {
                    "value": round(len(recs2) / cdt2, 1), "unit": "keypoints/s", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
                    "same_records_as_1_thread": bool(len(recs2) == len(recs) and recs2.tobytes() == recs.tobytes()),
                    "sample": "the same restatement built with -fopenmp (blur / DoG / subsample / detect / per-keypoint / descriptor loops over %s threads), same %d^3 volume: %d records in %.2f s (blur %.2f s, DoG %.2f s, detect %.2f s, keypoints %.2f s, descriptors %.2f s)"
                              % (os.environ["OMP_NUM_THREADS"], m, len(recs2), cdt2, st2.t_blur, st2.t_dog, st2.t_detect, st2.t_features, st2.t_desc)}
            except OSError as e:   # the OpenMP library is an extra: its absence must not cost the line
                out["cpu_baseline_ncores"] = {"value": None, "note": "OpenMP build of the oracle not available: %r" % (e,)}
    expect = feats.copy() if (rank == 0 and world > 1) else None
    ctx.close()
    del dvol
    if dist is not None:
        phase("volumes: leaving the process group")
        dist.barrier()
        dist.destroy_process_group()   # the measurement is complete; ranks other than 0 are done and leave the GPUs
    if wd is not None:
        wd.done()
    if rank != 0:
        return
    if world == 1 or args.mode == "volumes":
        print(json.dumps(out), flush=True)
        return
    # ---- N > 1: the line's value is the Z-slab split of ONE volume over the N GPUs (strong scaling) ----
    torch.cuda.empty_cache()
    vol_out = {k: out[k] for k in ("value", "ms_per_step", "scaling")}
    # The one-process C driver first (no launcher, no rendezvous, peer copies before RCCL): whatever happens to the per-process job
    # afterwards -- its nccl backend has only ever run over gloo on one GPU -- one lease of a node always yields a strong-scaling number.
    zc = zslab_c_child(args, world, expect, args.zslab_limit)
    z = zslab_child(args, world, expect, args.zslab_limit)
    ok = z.get("status") == "ok" and z.get("value") is not None
    value_source = "zslab: one process per GPU, halos over torch.distributed (%s)" % ("nccl = RCCL" if os.environ.get("SIFT3D_DIST_BACKEND", "nccl") == "nccl" else os.environ.get("SIFT3D_DIST_BACKEND"))
    if not ok:   # the C driver's result stands in, and the line says so
        for tr in ("peer_copy", "rccl"):
            c = zc.get(tr) or {}
            if c.get("ms_per_step") and c.get("records") and c.get("same_bytes_as_single_gpu") is not False:
                z = dict(z, value=round(c["records"] / (c["ms_per_step"] * 1e-3), 1), ms_per_step=c["ms_per_step"],
                         same_bytes_as_single_gpu=c.get("same_bytes_as_single_gpu"),
                         workload="ONE %s float32 blob-field volume cut into %d Z-slabs, full featExtract path, all octaves" % (vol_label, world),
                         parallelism="zslab%d: ONE process drives all %d devices from C (sift3d_zslab_extract_resident), halos by %s"
                                     % (world, world, "peer copies" if tr == "peer_copy" else "RCCL send/recv"),
                         records_to_rank0="the ranks' descriptor kernels store into one pinned list (C driver)",
                         per_process_job={"status": z.get("status"), "exit_code": z.get("exit_code"), "stderr_tail": z.get("stderr_tail")})
                value_source = "zslab_c/%s: the per-process job failed (%s), the one-process C driver's measurement of the same split stands in" % (tr, z.get("status"))
                ok = True
                break
    out["volumes_value"] = vol_out["value"] if replicas else None
    out["volumes_ms_per_step"] = vol_out["ms_per_step"]
    out["volumes_scaling"] = "weak" if replicas else None
    out["volumes_note"] = ("every rank extracting its own %s volume, barrier to barrier, max over ranks (rounds 1-4's headline)" % vol_label
                           if replicas else "rank 0 alone extracting the whole %s volume: the single-GPU time and records the Z-slab run is "
                                            "compared with (no replica leg for --config runs)" % vol_label)
    out["value"] = z.get("value") if ok else None
    out["ms_per_step"] = z.get("ms_per_step") if ok else None
    out["scaling"] = "strong"
    out["same_bytes_as_single_gpu"] = z.get("same_bytes_as_single_gpu")
    out["speedup_vs_single_gpu"] = round(vol_out["ms_per_step"] / z["ms_per_step"], 3) if ok and z.get("ms_per_step") else None
    out["config"]["workload"] = z.get("workload") or ("ONE %s float32 blob-field volume cut into %d Z-slabs (the child job failed: see `zslab`)" % (vol_label, world))
    out["config"]["parallelism"] = z.get("parallelism") or "zslab%d" % world
    out["config"]["records_to_rank0"] = z.get("records_to_rank0")
    out["config"]["single_gpu_workload"] = "%s volume on one GPU (`volumes_*`; `roofline`, `pyramid`, `stages` are rank 0's single-GPU kernels)" % vol_label
    out["value_source"] = value_source if ok else None
    out["zslab"] = z
    out["zslab_c"] = zc
    # (kept for readers of the rounds 3-4 line)
    out["zslab_value"], out["zslab_ms_per_step"], out["zslab_same_bytes_as_single_gpu"] = z.get("value"), z.get("ms_per_step"), z.get("same_bytes_as_single_gpu")
    print(json.dumps(out), flush=True)
    if not ok:
        sys.stderr.write("bench.py: neither the per-process Z-slab job (%s) nor the one-process C driver (%s) produced a result: value is null, exit code 5\n"
                         % (z.get("status"), zc.get("status")))
        sys.exit(5)


if __name__ == "__main__":
    main()
