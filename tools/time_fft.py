"""Ring Fourier stage time of one 8-component spin-0 map2alm (HX_LIBRARY selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
if os.environ.get("FFT_CAP"):  # rings whose Bluestein convolution is longer run as two half-length passes (k_ring_subdft_split)
    hx._lib.check(hx._lib.load().hx_set_max_lds_fft(int(os.environ["FFT_CAP"])))
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
plan = hx.Plan(nside, lmax)
nc = int(os.environ.get("NCOMP", 8))
m = torch.randn((nc, 12 * nside * nside), dtype=torch.float64, device="cuda")
pwm = os.environ.get("PW", "1")  # 0: no pixel weights; 1: an array with the symmetry of healpy's weights (short path); 2: random weights (generic path)
pw = None if pwm == "0" else (torch.ones(12 * nside * nside, dtype=torch.float64, device="cuda") if pwm == "1" else 1.0 + 0.01 * torch.rand(12 * nside * nside, dtype=torch.float64, device="cuda"))
plan.map2alm(m, 0, pix_weights=pw)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.map2alm(m, 0, pix_weights=pw)
print(os.environ.get("HX_LIBRARY", "default").split("/")[-1], "ring_fft ms (%d comps%s):" % (nc, ", pixel weights" if pw is not None else ""), round(hx._lib.profile_get("ring_fft")[1], 2))
