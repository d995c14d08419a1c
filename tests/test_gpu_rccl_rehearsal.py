"""The N > 1 branch of bench.py (RCCL scatter of the inputs, gather of the mels, barrier, max-over-ranks all-reduce) on real
hardware: a gpurun box has ONE GPU, so the group has a single rank, but every collective is issued through torch.distributed's
"nccl" (= RCCL) backend with the tensors the 8-GPU run uses (fp32 units, int64 speaker ids, fp64 timing scalar).  The
multi-rank logic itself is covered on CPU with gloo (tests/test_cpu_shard.py, tests/test_cpu_bench_plumbing.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,batch,frames", [("sampler", 3, 64), ("e2e", 2, 32), ("full_tts", 2, 16)])
def test_bench_single_rank_rccl_group(workload, batch, frames):
    """every workload of bench.py through the RCCL branch: sampler (configs[1] / [2]), e2e (configs[3]: the waveform is gathered too),
    full_tts (configs[4]: phones / tones scattered as int64, mel and waveform gathered)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-rccl", "--workload", workload, "--steps", "1", "--warmup", "1", "--nfe", "2",
                        "--batch", str(batch), "--frames", str(frames), "--no-extras", "--no-cpu-baseline", "--no-profile"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["metric"] == "mel_frames_per_sec" and line["value"] > 0
    assert line["config"]["utterances_per_gpu"] == batch and line["config"]["frames"] == frames and line["config"]["pipeline"] == workload
    assert line["config"]["gathered"] == (["mel"] if workload == "sampler" else ["mel", "wav"])
