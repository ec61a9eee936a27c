"""Time the SR-head gradient (sr_grad.SRImage: forward that keeps its activations + backward to the 128^2 x 32 feature image) at the
FFHQ head size, per view.    python tools/time_sr_grad.py [views]
Round 4: before / after nfe_bias_act_backward replaced the element-wise torch kernels between the backward-data convolutions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerffaceediting_amd import sr_grad  # noqa: E402
if os.environ.get("SR_GRAD_OLD"):            # A/B: an older copy of the module placed beside it
    from nerffaceediting_amd import _sr_grad_old as sr_grad  # noqa: E402,F811
from nerffaceediting_amd.training.superresolution import SuperresolutionHybrid8XDC  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True, channel_base=32768, channel_max=512,
                                   fused_modconv_default="inference_only").to(dev)
    feat = torch.randn(N, 128, 128, 32, device=dev)
    ws = torch.randn(N, 14, 512, device=dev)
    g = torch.randn(N, 512, 512, 3, device=dev)

    def timeit(fn, it=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it

    img, saved = sr_grad.sr_forward_saving(sr, feat, ws, "const")
    t_f = timeit(lambda: sr_grad.sr_forward_saving(sr, feat, ws, "const"))
    t_b = timeit(lambda: sr_grad.sr_backward(sr, saved, g))
    print(f"SR head gradient, {N} view(s) 128^2 x 32 -> 512^2: forward keeping activations {t_f:.3f} ms, backward {t_b:.3f} ms")


if __name__ == "__main__":
    main()
