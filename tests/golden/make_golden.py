#!/usr/bin/env python3
"""Regenerates tests/golden/ in the BUILD container (needs /root/reference).

Fixtures are data only (inputs are regenerated from the seeded blob-field
generator, so only expected outputs are stored):

  refbin_blob64.key, refbin_blob128.key.gz
      .key files written by the CPU featExtract binary that the reference
      repository ships (R/bin/Linux/featExtract, an upstream CPU-only build of
      the same pipeline), run here on blob-field volumes written by our NIfTI
      writer.  The binary is executed unprivileged (uid nobody, no new privs)
      through an inherited file descriptor; nothing of it is copied.
  refbin_aniso_w.key, refbin_aniso_ws.key
      the same binary with -w / -ws on the anisotropic WORLD_CASE volume of
      tests/_oracle.py (isotropic resampling + qto_xyz / sto_xyz world coordinates).
  oracle_aniso_w.key, oracle_aniso_ws.key
      the oracle CLI's output for the same two runs (regression pins).
  ref_taps.json
      raw Gaussian taps (hex floats) produced by the reference's own
      GaussianMask.cpp (compiled as it lies into oracle/_ref) for every sigma the
      path uses.
  oracle_blob64_{sift,brief,rrief,nrrief}.key, oracle_blob80x64x48_sift.key
      the oracle's own output (regression pins; the -b* variants exist only as
      commented alternatives in the reference, so nothing of the reference can
      produce them).
  ref_counts.json
      record counts of the source-built reference CPU path measured during the
      survey for the same generator.
"""
import ctypes as C
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle

REFBIN = "/root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/bin/Linux/featExtract"
CLI = _oracle.CLI


def run_refbin(nii, key, cwd, flags=()):
    with open(REFBIN, "rb") as f:
        fd = f.fileno()
        cmd = ["timeout", "300", "setpriv", "--reuid=65534", "--regid=65534", "--clear-groups", "--no-new-privs",
               "/lib64/ld-linux-x86-64.so.2", "/dev/fd/%d" % fd] + list(flags) + [nii, key]
        subprocess.run(cmd, cwd=cwd, check=True, pass_fds=(fd,), stdout=subprocess.DEVNULL)


def main():
    _oracle.build()
    work = os.path.join(ROOT, "gpurun_out", "golden_work")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    os.chmod(work, 0o777)
    for n in (64, 128):
        nii = os.path.join(work, "blob%d.nii" % n)
        subprocess.run([CLI, "--synth", str(n), str(n), str(n), "12345", nii], check=True)
        os.chmod(nii, 0o644)
        run_refbin("blob%d.nii" % n, "ref%d.key" % n, work)
    shutil.copy(os.path.join(work, "ref64.key"), os.path.join(HERE, "refbin_blob64.key"))
    with open(os.path.join(work, "ref128.key"), "rb") as f, gzip.GzipFile(os.path.join(HERE, "refbin_blob128.key.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    for mode, flag in (("sift", None), ("brief", "-b"), ("rrief", "-br"), ("nrrief", "-bn")):
        cmd = [CLI] + ([flag] if flag else []) + [os.path.join(work, "blob64.nii"), os.path.join(HERE, "oracle_blob64_%s.key" % mode)]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    nii = os.path.join(work, "blob80.nii")
    subprocess.run([CLI, "--synth", "80", "64", "48", "777", nii], check=True)
    subprocess.run([CLI, nii, os.path.join(HERE, "oracle_blob80x64x48_sift.key")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    nii = os.path.join(work, "aniso.nii")
    subprocess.run(_oracle.world_case_args(nii), check=True)
    os.chmod(nii, 0o644)
    for flag in ("-w", "-ws"):
        run_refbin("aniso.nii", "ref_aniso%s.key" % flag, work, [flag])
        shutil.copy(os.path.join(work, "ref_aniso%s.key" % flag), os.path.join(HERE, "refbin_aniso_%s.key" % flag[1:]))
        subprocess.run([CLI, flag, nii, os.path.join(HERE, "oracle_aniso_%s.key" % flag[1:])], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    ref = _oracle.load_ref()
    assert ref is not None, "oracle/_ref missing"
    k = np.float32(2.0 ** (1.0 / 3.0)); s = np.float32(1.6)
    sig = {"init": np.sqrt(np.float32(1.6) * np.float32(1.6) - np.float32(0.5) * np.float32(0.5)),
           "init_2x": np.sqrt(np.float32(1.6) * np.float32(1.6) - np.float32(1.0) * np.float32(1.0))}
    for j in range(1, 6):
        sig["level%d" % j] = np.float32(s * np.sqrt(np.float32(k * k - np.float32(1)))); s = np.float32(s * k)
    sig["ori_hist"] = np.float32(0.5); sig["brief"] = np.float32(0.95)
    out = {}
    for name, sg in sig.items():
        n = ref.ref_gauss_filter_size(float(sg), 0.01)
        t = np.zeros(n, np.float32)
        ref.ref_gauss_taps_raw(float(sg), n, t.ctypes.data)
        out[name] = {"sigma_hex": float(sg).hex(), "ntaps": int(n), "raw_taps_hex": [float(v).hex() for v in t]}
    json.dump(out, open(os.path.join(HERE, "ref_taps.json"), "w"), indent=1)
    json.dump({"generator": "sift3d_synth_blobs seed 12345", "source": "SURVEY.md section 6 (reference CPU path built from source during the survey)",
               "records": {"64": 74, "128": 1698, "256": 19216}}, open(os.path.join(HERE, "ref_counts.json"), "w"), indent=1)
    shutil.rmtree(work, ignore_errors=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
