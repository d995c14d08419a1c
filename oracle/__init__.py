"""CPU oracle: a numpy restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under latent-diffusion-speech_amd/ may import this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
only as the checker / reported CPU baseline, never as the thing measured or shipped.

Parity pinning: the reference holds no tests or golden vectors, so every
function here is pinned against outputs of the reference's own runnable leaf modules
imported in the build container (tests/golden/make_fixtures.py -> tests/golden/*.npz),
checked by tests/test_oracle_vs_golden.py.

Modules:
  schedule.py  GaussianDiffusion buffers + NoiseScheduleVP      (reference diffusion/diffusion.py:46-87,
                                                                 diffusion/dpm_solver_pytorch.py:6-167,1253-1292)
  solvers.py   DPM-Solver++(2M), UniPC-bh2, DDPM, DDIM, PLMS    (dpm_solver_pytorch.py:433-1213, uni_pc.py:471-672,
                                                                 diffusion.py:95-167,189-343)
  unet1d.py    UNet1DConditionModel forward                     (diffusion/unet1d/*.py, see function docstrings)
  vocoder.py   HiFi-VAEGAN Generator forward                    (encoder/hifi_vaegan/modules/models.py:161-272)
  unit2mel.py  Unit2Mel front end + end-to-end pipeline         (diffusion/unit2mel.py:51-88, diffusion/vocoder.py:32-33)
"""
