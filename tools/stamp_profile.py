"""Stamp counter summaries copied into profiles/ with what they were taken on, so that bench.py can tell whether a committed summary
still describes the kernels of the running tree:

    python tools/stamp_profile.py profiles/r03_hbm_traffic.json profiles/r03_mfma_util.json ...

adds {"commit": <HEAD at stamping time>, "lib_sha256": <liblds.so>, "src_sha256": {csrc file: hash}} to each JSON.  The GPU box has no
.git, which is why the stamp is applied here, in the container, to files merged back from gpurun_out/ -- right after the run, before
anything under csrc/ changes.  bench.py marks a summary `stale` when a source file of the kernel's family differs from the stamp."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "latent-diffusion-speech_amd", "csrc")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def stamp():
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "latent-diffusion-speech_amd/csrc"], capture_output=True, text=True).stdout.strip())
    srcs = {os.path.basename(f): sha(f) for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))}
    lib = os.path.join(ROOT, "latent-diffusion-speech_amd", "lds", "liblds.so")
    return {"commit": commit, "csrc_dirty_at_stamp": dirty, "lib_sha256": sha(lib) if os.path.exists(lib) else None, "src_sha256": srcs}


def main():
    st = stamp()
    for f in sys.argv[1:]:
        d = json.load(open(f))
        d.update(st)
        json.dump(d, open(f, "w"), indent=1)
        print("stamped", f, st["commit"][:10] if st["commit"] else None)


if __name__ == "__main__":
    main()
