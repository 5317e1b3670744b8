"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32) of the reference's interleaved text+latent
training/sampling hot path, used only as the CHECKER:

  * `tests/`                      (parity tests, golden-vector tests)
  * `__graft_entry__.smoke()`     (one tiny parity check on cuda:0)
  * `bench.py` `cpu_baseline` leg (timed "port" baseline on the host cores)

Nothing under `transfusion_pytorch_amd/` may import this package: the product
path runs on the HIP C-ABI library and fails loudly when it is missing.

Parity status: the restatement is PINNED against the unmodified reference
(`/root/reference`, imported through `oracle/shims`) by
`oracle/make_golden.py`, which wrote `tests/golden/*.pt` from the REFERENCE's
outputs; `tests/test_oracle_golden.py` checks the restatement against them.
Third-party arithmetic the reference imports but does not vendor (RoPE from
`rotary_embedding_torch`, midpoint rule from `torchdiffeq`) is restated from
the packages' public semantics in `oracle/shims` — "parity unpinned" for those
two pieces (no reference test pins their absolute values).
"""
