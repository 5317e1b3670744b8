"""one `sample_many` run of SURVEY 8(d) config 5 (forced modality at the start) for a rocprofv3 kernel trace of the decode loop.
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o p -- python $R/tools/prof_sample.py [max_length])
    python tools/prof_summary.py /tmp/ps/p_kernel_trace.csv --steps <global steps printed by TFX_SAMPLE_TIMING=1>"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import sample_prompts                      # noqa: E402
from transfusion_pytorch_amd import Transfusion      # noqa: E402

max_len = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=1024, depth=24)).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1234)
prompts = sample_prompts(16, dev, g)
noise = torch.randn(16, 384, device=dev, generator=g)
kw = dict(max_length=max_len, modality_steps=16, cfg_scale=3., text_temperature=0., init_modality_noise=noise, fixed_modality_shape=(4,), force_modality_at_start=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
m.sample_many(prompts, **kw)
torch.cuda.synchronize()
print(f'sample_many(max_length={max_len}): {time.perf_counter() - t0:.3f} s', file=sys.stderr)
