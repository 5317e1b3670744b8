"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol include/tfx.h declares, the
host packer reproduces the reference's packed layout (via the oracle's restatement), and the product path refuses
to run without the HIP device."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle.cases import build_case
from oracle.transfusion_oracle import kv_end_from_positions, pack_batch, rotary_positions
from transfusion_pytorch_amd import Transfusion, capi
from transfusion_pytorch_amd.packing import fast_signature, scan_batch, scan_signature, token_maps, token_segments


def test_c_abi_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), 'run `python -m transfusion_pytorch_amd.build` (hipcc cross-compiles gfx950 without a GPU)'
    lib = ctypes.CDLL(capi.LIB_PATH)
    assert len(capi.FUNCTIONS) >= 30
    for name in capi.FUNCTIONS:
        assert hasattr(lib, name), f'{name} declared in include/tfx.h but not exported by libtfx_hip.so'
    assert capi.lib().tfx_version().startswith(b'tfx-hip gfx950')
    # every struct of the header became a ctypes Structure with pointer-aligned layout
    for name, S in capi.STRUCTS.items():
        assert ctypes.sizeof(S) % 8 == 0 or all(t is not ctypes.c_void_p for _, t in S._fields_), name


def test_launch_list_image_and_runner_error_paths():
    """host logic of the launch-list replay (no kernel is launched): the native image mirrors the Python list (op codes, args
    by address, positional arguments packed in declaration order) and tfx_run_list reports the failing index."""
    from transfusion_pytorch_amd.engine import LaunchList, raw_args
    lib = capi.lib()
    L = LaunchList()
    a = capi.make_args('tfx_gemm_nt_args', M=1, N=2, K=3)
    L.append(('tfx_gemm_nt', a))
    live = [16, 32, None, 48, 64, 5, 6, 0.25]
    L.append((lib.tfx_output_to_flow, live))
    L.append((lib.tfx_scatter_rows_bf16, (8, 1, 2, 24, 3, None, 4)))
    arr = L.native()
    assert [arr[k].op for k in range(3)] == [capi.ENUMS['TFX_OP_GEMM_NT'], capi.ENUMS['TFX_OP_OUTPUT_TO_FLOW'], capi.ENUMS['TFX_OP_SCATTER_ROWS_BF16']]
    assert arr[0].args == ctypes.addressof(a)
    R = capi.STRUCTS['tfx_raw_args']
    r1 = R.from_address(arr[1].args); r2 = R.from_address(arr[2].args)
    assert (r1.p0, r1.p1, r1.p2, r1.p3, r1.p4, r1.i0, r1.i1, r1.f0) == (16, 32, None, 48, 64, 5, 6, 0.25)
    assert (r2.p0, r2.i0, r2.i1, r2.p1, r2.i2, r2.p2, r2.i3) == (8, 1, 2, 24, 3, None, 4)
    live[2] = 128                                  # a mutable positional list is re-packed on the next replay
    assert L.native() is arr and R.from_address(arr[1].args).p2 == 128
    L.append(('tfx_gemm_tn', capi.make_args('tfx_gemm_tn_args')))
    assert len(L.native()) == 4                    # appended launches rebuild the image
    assert raw_args('tfx_colsum_f32', (8, 1, 2, 3, 16)).p1 == 16
    # stream tags: Side items run on the library's side stream, fork / join items carry their event slot
    from transfusion_pytorch_amd.engine import Side
    L2 = LaunchList([('tfx_gemm_nt', a), ('tfx_fork', 5), Side(('tfx_gemm_tn', capi.make_args('tfx_gemm_tn_args'))), ('tfx_join_record', 2),
                     ('tfx_join_wait', 2), ('tfx_join', 63)])
    arr2 = L2.native()
    assert [arr2[k].stream for k in range(6)] == [0, 5, 1, 2, 2, 63]
    assert [arr2[k].op for k in range(6)] == [capi.ENUMS[n] for n in ('TFX_OP_GEMM_NT', 'TFX_OP_FORK', 'TFX_OP_GEMM_TN', 'TFX_OP_JOIN_RECORD',
                                                                       'TFX_OP_JOIN_WAIT', 'TFX_OP_JOIN')]
    assert all(arr2[k].args for k in range(6))
    assert lib.tfx_set_single_stream(1) == 0 and lib.tfx_set_single_stream(0) == 1
    # runner: empty list is a no-op, unknown op / NULL args stop at their index
    failed = ctypes.c_int32(-1)
    assert lib.tfx_run_list(None, 0, None, ctypes.byref(failed)) == 0
    bad = (capi.STRUCTS['tfx_launch'] * 2)()
    bad[0].op = 999; bad[0].args = ctypes.addressof(a)
    assert lib.tfx_run_list(bad, 1, None, ctypes.byref(failed)) == -100 and failed.value == 0
    bad[0].op = capi.ENUMS['TFX_OP_GEMM_NT']; bad[0].args = None
    assert lib.tfx_run_list(bad, 1, None, ctypes.byref(failed)) == -2 and failed.value == 0
    assert lib.tfx_run_list(None, 3, None, ctypes.byref(failed)) == -1


def _native(cfg):
    dl = cfg.dim_latents if len(cfg.dim_latents) > 1 else cfg.dim_latents[0]
    return Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=dl,
                       transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))


def test_launch_list_fingerprint_follows_everything_a_capture_freezes():
    """`tfx_list_fingerprint` (host logic, no device): the hash a training-graph replay is validated with (LaunchList.replay_auto) moves when an args
    struct field, a positional argument, a host struct behind a two-struct entry point, an op, a stream tag or the single-stream switch changes -
    and only then."""
    from transfusion_pytorch_amd.engine import LaunchList, Side
    lib = capi.lib()
    L = LaunchList()
    a = capi.make_args('tfx_gemm_nt_args', M=64, N=128, K=256)
    post = capi.make_args('tfx_adaln_post_args', T=7, d=64)
    pre = capi.make_args('tfx_adaln_pre_args', T=7, d=64)
    live = [16, 32, None, 48, 64, 5, 6, 0.25]
    L.append(('tfx_gemm_nt', a))
    L.append(Side(('tfx_gemm_nt', a)))
    L.append((lib.tfx_output_to_flow, live))
    L.append((lib.tfx_adaln_post_pre_fwd, (ctypes.addressof(post), ctypes.addressof(pre))))
    L.append(('tfx_fork', 3))
    keep = (post, pre)

    def fp(lo=0, hi=None):
        arr = L.native()
        out = ctypes.c_int64()
        hi_ = len(L) if hi is None else hi
        from transfusion_pytorch_amd.engine import _LAUNCH
        assert lib.tfx_list_fingerprint(ctypes.byref(arr, lo * ctypes.sizeof(_LAUNCH)), hi_ - lo, ctypes.byref(out)) == 0
        return out.value

    base = fp()
    assert fp() == base and fp(0, 2) != base
    a.M = 65; f1 = fp(); assert f1 != base
    a.M = 64; assert fp() == base
    live[7] = 0.5; assert fp() != base                       # a positional (re-packed) argument
    live[7] = 0.25; assert fp() == base
    post.T = 8; assert fp() != base                          # a host struct the entry point only points at
    post.T = 7; assert fp() == base
    prev = lib.tfx_set_single_stream(1)
    try:
        assert fp() != base
    finally:
        lib.tfx_set_single_stream(prev)
    assert fp() == base
    assert lib.tfx_list_fingerprint(None, 1, ctypes.byref(ctypes.c_int64())) == -1


@pytest.mark.parametrize('name', ['tiny1', 'small2', 'mid2', 'head8', 'canon512'])
def test_packer_matches_reference_layout(name):
    cfg, sd, batch, times, noise = build_case(name)
    m = _native(cfg)
    assert set(m.state_dict()) == set(cfg.state_dict_shapes())
    for k, shp in cfg.state_dict_shapes().items():
        assert tuple(m.state_dict()[k].shape) == tuple(shp), k
    P = m._scan(batch, add_sos_eos=True)
    O = pack_batch(cfg, batch)                                  # oracle restatement of MP:206-377 (pinned to the reference)
    assert P.positions == O.positions and P.total_tokens == O.total_tokens and P.n_full == O.text.shape[1]
    text = P.text_host.copy().reshape(-1)
    text[P.text_dest] = torch.cat([t.reshape(-1) for t in P.user_text]).numpy()
    assert np.array_equal(text.reshape(O.text.shape), O.text.numpy())
    n = P.n_full - 1
    tm = token_maps(P, n, cfg.num_modalities)
    assert np.array_equal(tm.kv_end, kv_end_from_positions(O.positions, P.b, n).numpy())
    assert np.array_equal(tm.rot_pos, rotary_positions(O.positions, P.b, n).numpy())
    # q_start is the inverse view of kv_end: key j is visible to query i  <=>  i >= q_start[j]
    i = np.arange(n)
    for bi in range(P.b):
        vis_q = i[None, :] < tm.kv_end[bi][:, None]            # [query, key]
        vis_k = i[:, None] >= tm.q_start[bi][None, :]
        assert np.array_equal(vis_q, vis_k)
    ss, sl = token_segments(tm.tok_inst)
    assert sl.sum() == P.b * n and (sl > 0).all()
    flat = tm.tok_inst.reshape(-1)
    for s0, l0 in zip(ss[:200], sl[:200]):
        assert (flat[s0:s0 + l0] == flat[s0]).all()
    sig, texts, lats = fast_signature(batch)
    assert len(texts) == len(P.user_text) and {t: len(v) for t, v in lats.items()} == {t: len(v) for t, v in P.latents.items()}


def test_state_dict_roundtrip_and_flat_views():
    cfg, sd, batch, times, noise = build_case('tiny1')
    m = _native(cfg)
    m.load_state_dict(sd, strict=True)
    for k, v in sd.items():
        assert torch.equal(m.state_dict()[k], v), k
    # parameters are views into ONE flat buffer
    k = 'transformer.layers.1.2.fn.net.0.weight'
    m.store.view(k).add_(1.0)
    assert torch.equal(m.state_dict()[k], sd[k] + 1.0)


def test_unsupported_options_raise_and_no_cpu_fallback():
    with pytest.raises(AssertionError):
        Transfusion(num_text_tokens=8, dim_latent=16, transformer=dict(dim=64, depth=1, heads=1), modality_processing='nope')
    with pytest.raises(NotImplementedError):
        Transfusion(num_text_tokens=8, dim_latent=16, transformer=dict(dim=64, depth=1, heads=1, dim_head=128))   # kernels hold 64 columns per head
    Transfusion(num_text_tokens=8, dim_latent=16, transformer=dict(dim=64, depth=1, heads=2, dim_head=8, use_flex_attn=True))   # small heads run zero-padded; flex = backend name only
    assert Transfusion(num_text_tokens=8, dim_latent=16, reconstruction_loss_weight=0.1, transformer=dict(dim=64, depth=1, heads=1)).has_recon_loss
    with pytest.raises(NotImplementedError):
        Transfusion(num_text_tokens=8, dim_latent=16, transformer=dict(dim=64, depth=1, heads=1, dropout=0.1))
    with pytest.raises(AssertionError):                   # T:1396: the positional embedding needs the number of axial dimensions
        Transfusion(num_text_tokens=8, dim_latent=16, add_pos_emb=True, transformer=dict(dim=64, depth=1, heads=1))
    m = Transfusion(num_text_tokens=8, dim_latent=16, transformer=dict(dim=64, depth=1, heads=1))
    with pytest.raises(capi.TfxError):
        m([[torch.randint(0, 8, (4,)), torch.randn(2, 16)]])


def test_bench_workload_matches_the_survey_definition():
    """bench.py's synthetic sample and flop count are the ones SURVEY.md section 8(d) defines: 64 parts alternating 24 text tokens /
    a (4, 384) latent pack to 1025 tokens (n = 1024 after the last-token drop), F_core = 180.6 / 1624.5 GFLOP per sample."""
    import bench
    assert abs(bench.f_core_per_sample(512, 8) / 1e9 - 180.6) < 0.05
    assert abs(bench.f_core_per_sample(1024, 24) / 1e9 - 1624.5) < 0.05
    gen = torch.Generator().manual_seed(1234)
    batch = bench.canonical_batch(2, 'cpu', gen)
    assert len(batch[0]) == 64 and batch[0][1].shape == (4, 384) and batch[0][-2].numel() == 23
    m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=64, depth=1, heads=1))
    P = scan_batch(batch, num_modalities=1, dim_latents=(384,), sos_id=m.sos_id, eos_id=m.eos_id, meta_id=m.meta_id, som_ids=m.som_ids,
                   eom_ids=m.eom_ids, add_sos_eos=True)
    assert P.n_full == 1025 and P.total_tokens == 2 * 1025 and P.positions[0][0] == (0, 28, 4)
    # BASELINE config 4 (`bench.py --config 4`): two modality types, even instances (0, (4,384)), odd ones (1, (2,192)), text fillers 25 (last 24):
    # packs to 1025 tokens with positions [(0,29,4), (1,62,2), (0,93,4), ...] (the survey's probe), V = 392; mask-aware F_core = 670.7 GFLOP per sample
    batch4 = bench.two_modality_batch(2, 'cpu', gen)
    assert len(batch4[0]) == 64 and batch4[0][1][0] == 0 and batch4[0][1][1].shape == (4, 384) and batch4[0][3][0] == 1 and batch4[0][3][1].shape == (2, 192)
    m4 = bench.build_model(64, 1, True)
    assert m4.md.vocab == 392
    P4 = scan_batch(batch4, num_modalities=2, dim_latents=(384, 192), sos_id=m4.sos_id, eos_id=m4.eos_id, meta_id=m4.meta_id, som_ids=m4.som_ids,
                    eom_ids=m4.eom_ids, add_sos_eos=True)
    assert P4.n_full == 1025 and P4.total_tokens == 2 * 1025 and list(P4.positions[0][:3]) == [(0, 29, 4), (1, 62, 2), (0, 93, 4)]
    assert abs(bench.f_core_per_sample(768, 16, inst_lens=bench.inst_lens_of(True)) / 1e9 - 670.7) < 0.05
    assert bench.CONFIGS[4] == dict(dim=768, depth=16, two=True) and bench.CONFIGS[3] == dict(dim=1024, depth=24, two=False)


def test_default_times_match_reference_golden():
    """`Transfusion._default_times` (host glue in front of the kernels; default_modality_length_to_time_fn, T:186-200) fed with the uniform
    draws of tests/golden/cfg1.pt must give the UNMODIFIED reference's times exactly (the GPU test repeats this through forward())."""
    import os
    import numpy as np
    import torch
    from oracle.detdata import count_instances
    from oracle.make_golden_cfg import cfg_case, patched_uniforms
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'cfg1.pt'), weights_only=False)
    cfg, sd, batch, noise, draws = cfg_case()
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    counts = np.asarray(count_instances(batch))
    with patched_uniforms([draws['u_k'], draws['u_t']]) as pu:
        times = model._default_times(counts)
        assert not pu.queue
    assert torch.equal(times.cpu(), g['times'])
    assert model._default_times(np.zeros(3, dtype=np.int64)).shape == (3, 0)          # T:189-190


def test_axial_positional_embedding_and_external_encoder_wiring():
    """host side of `add_pos_emb` / `pre_post_transformer_enc_dec` (T:1384-1403, T:1451-1494): the embedding of a position is the sum of its
    per-axis vectors, row-major over the axes; factorised evaluation at the maximum extents + slicing gives the same rows (MP:1003-1045); the user's
    encoder / decoder take the reference's attribute names (so its checkpoints load), own no entry in the flat buffer, and the packer sees the
    PROJECTED axial shape (MP:738-741)."""
    from torch import nn
    from transfusion_pytorch_amd.axial import ContinuousAxialPositionalEmbedding
    pe = ContinuousAxialPositionalEmbedding(dim=32, num_axial_dims=2)
    rows = pe((3, 4), flatten=True)
    fac = pe((5, 6), return_factorized=True)
    assert rows.shape == (12, 32) and [tuple(f.shape) for f in fac] == [(5, 32), (6, 32)]
    assert torch.allclose(rows, pe.combine_factorized(fac, (3, 4), flatten=True), atol=1e-6)
    assert torch.allclose(rows.view(3, 4, 32)[2, 1], fac[0][2] + fac[1][1], atol=1e-6)
    with pytest.raises(AssertionError):
        pe((3, 4, 5))
    enc, dec = nn.Conv2d(4, 64, 3, 2, 1), nn.ConvTranspose2d(64, 4, 3, 2, 1, output_padding=1)
    m = Transfusion(num_text_tokens=8, dim_latent=4, channel_first_latent=True, pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2,
                    modality_default_shape=(8, 8), transformer=dict(dim=64, depth=1, heads=1))
    keys = set(m.state_dict())
    assert {'latent_to_model_projs.0.0.weight', 'model_to_latent_projs.0.1.weight', 'pos_emb_mlp.0.mlps.1.0.weight'} <= keys       # Sequential(enc, Rearrange) / (Rearrange, dec)
    assert not any(k in m.store.offsets for k in keys if k.startswith(('latent_to_model_projs', 'model_to_latent_projs', 'pos_emb_mlp')))
    ext = m.external_parameters()
    assert len(ext) == 4 + 12 and all(p.requires_grad for p in ext)
    assert m.md.ext_types == (0,) and m.md.pos_types == (0,)
    batch = [[torch.randint(0, 8, (3,)), (0, torch.randn(4, 8, 8))], [(0, torch.randn(4, 4, 8))]]
    out, ctx = m._ext_preprocess(batch, torch.full((2, 1), 0.5), return_loss=True)
    assert [tuple(p[1].shape) for s in out for p in s if isinstance(p, tuple)] == [(4, 4, 4), (2, 4, 4)] and out[0][1][1].device.type == 'meta'
    assert ctx[0]['shape'] == [(4, 4), (2, 4)] and [tuple(t.shape) for t in ctx[0]['tok']] == [(16, 64), (8, 64)] and ctx[0]['tok'][0].requires_grad
    assert [tuple(f.shape) for f in ctx[0]['flow']] == [(4, 8, 8), (4, 4, 8)]
    P = m._scan(out, add_sos_eos=True)
    assert P.inst_len.tolist() == [16, 8] and P.inst_shape == [(4, 4), (2, 4)]
    # the [meta] shape string of the first instance spells the projected shape "4,4"
    row = P.text_host[0].tolist()
    i = row.index(m.meta_id)
    assert [c - m.meta_id - 1 for c in row[i + 1:i + 4]] == [ord('4'), ord(','), ord('4')]
    clone = m._clone_architecture()
    assert clone.latent_to_model_projs[0][0] is not m.latent_to_model_projs[0][0]


def test_packer_matches_oracle_on_random_ragged_batches():
    """the native packer assembles its per-token arrays from per-part runs with array operations: ragged random batches (two modality types of
    1 - 3 axial dims, bare float tensors = type 0, empty / adjacent text parts, samples without modalities) against the oracle's per-token
    packer (pinned to the reference, tests/test_oracle_golden.py) - ids, positions, mask bounds, rotary positions; segments cover every token once"""
    from oracle.transfusion_oracle import OracleConfig
    cfg = OracleConfig(num_text_tokens=200, dim=64, depth=1, dim_latents=(8, 16), heads=1, dim_head=64)
    g = torch.Generator().manual_seed(7)
    ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))
    shapes = [[(4,), (2, 3), (1,), (11,)], [(3, 3), (2, 2, 2), (5,), (1, 1)]]
    for trial in range(25):
        batch = []
        for _ in range(ri(1, 6)):
            s = []
            for _ in range(ri(0, 8)):
                kind = ri(0, 3)
                if kind == 0:
                    s.append(torch.randint(0, 200, (ri(0, 40),), generator=g))
                else:
                    ty = ri(0, 2)
                    x = torch.randn(*shapes[ty][ri(0, 4)], cfg.dim_latents[ty], generator=g)
                    s.append(x if (ty == 0 and kind == 2) else (ty, x))
            batch.append(s or [torch.randint(0, 200, (3,), generator=g)])
        O = pack_batch(cfg, batch)
        P = scan_batch(batch, num_modalities=2, dim_latents=cfg.dim_latents, sos_id=cfg.sos_id, eos_id=cfg.eos_id, meta_id=cfg.meta_id,
                       som_ids=cfg.som_ids, eom_ids=cfg.eom_ids, add_sos_eos=True)
        assert P.positions == O.positions and P.total_tokens == O.total_tokens and P.n_full == O.text.shape[1], trial
        # the product's scan runs on the structure signature alone (packing.scan_signature): field by field the same packed batch, both layouts
        for meta in (True, False):
            sig, tx, lat = fast_signature(batch)
            kw = dict(num_modalities=2, dim_latents=cfg.dim_latents, sos_id=cfg.sos_id, eos_id=cfg.eos_id, meta_id=cfg.meta_id, som_ids=cfg.som_ids,
                      eom_ids=cfg.eom_ids, add_sos_eos=meta, add_meta=meta)
            A, B = scan_batch(batch, **kw), scan_signature(sig, tx, lat, **kw)
            assert (A.b, A.n_full, A.total_tokens, A.positions, A.inst_shape) == (B.b, B.n_full, B.total_tokens, B.positions, B.inst_shape), trial
            for f in ('text_host', 'text_dest', 'cfg_droppable', 'inst_b', 'inst_m', 'inst_type', 'inst_off', 'inst_len', 'lens'):
                assert np.array_equal(getattr(A, f), getattr(B, f)), (trial, f)
            assert A.row_inst.keys() == B.row_inst.keys() == A.latents.keys() == B.latents.keys()
            for t in A.row_inst:
                assert np.array_equal(A.row_inst[t], B.row_inst[t]) and np.array_equal(A.row_pos[t], B.row_pos[t]) and len(A.latents[t]) == len(B.latents[t])
        text = P.text_host.copy().reshape(-1)
        if P.user_text:
            text[P.text_dest] = torch.cat([t.reshape(-1) for t in P.user_text]).numpy()
        assert np.array_equal(text.reshape(O.text.shape), O.text.numpy()), trial
        n = P.n_full - 1
        tm = token_maps(P, n, 2)
        assert np.array_equal(tm.kv_end, kv_end_from_positions(O.positions, P.b, n).numpy()), trial
        assert np.array_equal(tm.rot_pos, rotary_positions(O.positions, P.b, n).numpy()), trial
        ss, sl = token_segments(tm.tok_inst)
        assert sl.sum() == P.b * n and (sl > 0).all() and np.array_equal(ss, np.concatenate(([0], np.cumsum(sl)[:-1])))
        flat = tm.tok_inst.reshape(-1)
        for s0, l0 in zip(ss, sl):
            assert (flat[s0:s0 + l0] == flat[s0]).all() and (flat[s0] >= 0 or l0 <= 8) and s0 // n == (s0 + l0 - 1) // n


def test_reassigned_parameter_data_is_readopted_into_the_flat_buffer():
    """`p.data = w` moves a parameter out of the flat master buffer the kernels and the fused optimizer read: the version check copies it back into its
    slice, re-points the parameter and invalidates the bf16 shadows; `mark_weights_changed()` covers in-place edits through `.data` nothing can see."""
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=32, dim_latent=8, modality_default_shape=(2,), transformer=dict(dim=64, depth=1, dim_head=64, heads=1))
    st = m.store
    name, prm = next((n, p) for n, p in st.params.items() if p.dim() == 2)
    st._shadow_version = st.params_version()                       # "shadows are current"
    assert prm.data_ptr() == st.ptr(name)
    w = torch.randn_like(prm.data)
    prm.data = w                                                   # e.g. a checkpoint loader that assigns instead of copying
    assert prm.data_ptr() != st.ptr(name)
    st.params_version()
    assert prm.data_ptr() == st.ptr(name) and torch.equal(st.view(name), w) and st._shadow_version is None
    st._shadow_version = st.params_version()
    prm.data.mul_(0.5)                                             # invisible: no counter, no pointer
    assert st.params_version() == st._shadow_version
    m.mark_weights_changed()
    assert st._shadow_version is None
    with pytest.raises(ValueError):
        prm.data = torch.zeros(3, 3); st.params_version()


def test_ragged_count_buckets():
    """training-plan buckets for ragged instance / latent-row counts: steps of a quarter of the count's power of two, never below the minimum step,
    never below the count, and few distinct values over a corpus-like spread (a plan costs ~0.5 s to build; the cache holds 8)."""
    from transfusion_pytorch_amd.transfusion import _bucket
    assert [_bucket(x, 64) for x in (0, 1, 63, 64, 65, 200, 1024, 1025, 1950, 2048, 2049)] == [0, 64, 64, 64, 128, 256, 1024, 1280, 2048, 2048, 2560]
    assert [_bucket(x, 256) for x in (5, 256, 257, 7700, 8192, 8193)] == [256, 256, 512, 8192, 8192, 10240]
    for lo, hi, step in ((1700, 2100, 64), (7000, 8400, 256)):
        vals = {_bucket(x, step) for x in range(lo, hi)}
        assert all(b >= x for x in range(lo, hi) for b in [_bucket(x, step)]) and len(vals) <= 3


def test_weight_gradient_gemm_plan_rule():
    """`tfx_gemm_tn_plan` (host logic of the TN launcher, no device): 256 x 256 tiles from 8 tiles on, 128 x 128 below; the split count is the smallest
    that puts 0.9 x 256 (0.5 x 512) blocks on the chip in ONE round, chunks of at least 256 rows; explicit counts are honoured; grids are multiples of 8."""
    lib = capi.lib()

    def plan(M, N, K, splits=0, colsum=0, a_rowmap=0):
        a = capi.make_args('tfx_gemm_tn_args', M=M, N=N, K=K, lda=(N + 7) // 8 * 8, a_cols=(N + 7) // 8 * 8, ldb=(K + 7) // 8 * 8, b_cols=(K + 7) // 8 * 8,
                           ldc=K, k_valid=K, splits=splits, accumulate=1, alpha=1.0, colsum=colsum, a_rowmap=a_rowmap)
        out = [ctypes.c_int32(-9) for _ in range(4)]
        assert lib.tfx_gemm_tn_plan(ctypes.byref(a), *[ctypes.byref(o) for o in out]) == 0
        return tuple(o.value for o in out)                    # kind, tiles, splits, grid

    T = 65536
    # kind 3 (round 5) = the 256 x 256 tiling on the one-wave-per-SIMD kernel: same tiles, splits and grid as kind 2
    assert plan(T, 2816, 512) == (3, 22, 11, 248)              # 242 blocks on 256 CUs (8 splits = 176, 12 = a second round)
    assert plan(T, 1544, 512) == (3, 14, 17, 240)
    assert plan(T, 512, 1408) == (3, 12, 20, 240)
    assert plan(T, 512, 512) == (0, 16, 16, 256)               # 4 tiles of 256 x 256: the 128 x 128 form at half of its 512 block slots
    assert plan(T, 24576, 2048) == (3, 768, 1, 768)            # more tiles than slots: no split
    assert plan(T, 5632, 1024) == (3, 88, 2, 176)              # 3 splits would open a second round
    assert plan(512, 512, 512)[2] == 2                         # chunks stay >= 256 rows
    assert plan(2112, 512, 512) == (0, 16, 8, 128)             # 8 chunks of 320 rows (264 rounded up to 64): the 8th starts at 2240 > M - tn_block clamps it to an empty range
    assert plan(T, 1544, 512, splits=8) == (3, 14, 8, 112)     # explicit counts as given
    assert plan(1000, 200, 136)[0] == -1 and plan(T, 512, 512, a_rowmap=64)[0] == -1      # M % 64 != 0 / gathered rows: the register-staged kernel
    assert all(plan(T, n, k)[3] % 8 == 0 for n in (264, 520, 1544, 3080) for k in (384, 512, 768, 1024))


def test_nt_gemm_kernel_selection():
    """`tfx_gemm_nt_plan` (host logic of the NT launcher, no device): which kernel a shape runs on and its grid."""
    lib = capi.lib()
    FALLBACK, GLDS, MID, PP, SKINNY, DECODE, OW, OWP = range(8)

    def plan(M, N, K, epi='TFX_EPI_BF16', **kw):
        a = capi.make_args('tfx_gemm_nt_args', M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epi=capi.ENUMS[epi], **kw)
        kind, grid = ctypes.c_int32(-9), ctypes.c_int32(-9)
        rc = lib.tfx_gemm_nt_plan(ctypes.byref(a), ctypes.byref(kind), ctypes.byref(grid))
        return rc, kind.value, grid.value

    T = 65536
    # the config-2 training step's shape -> kernel map (round 6: the plan names the one-wave kernels)
    assert plan(T, 1544, 512) == (0, OWP, 256 * 7)             # [q | k | v | gates] projection: 256 x 256 tiles, ragged last N tile included; persistent one-wave kernel
    assert plan(T, 512, 512) == (0, OWP, 512)                  # out projection, dX of the projections: exactly two tiles per CU
    assert plan(T, 512, 1408) == (0, OWP, 512)                 # FeedForward down projection
    assert plan(T, 512, 128)[1] == OW                          # K < 192: the one-tile-per-block one-wave kernel (the persistent stream needs three K-tiles)
    assert plan(T, 2816, 512, 'TFX_EPI_GEGLU') == (0, PP, 256 * 11)       # GEGLU forward: fused epilogue -> ping-pong kernel (8 waves)
    assert plan(T, 1408, 512, 'TFX_EPI_GEGLU_BWD') == (0, PP, 256 * 6)    # GEGLU backward
    assert plan(T, 512, 512, 'TFX_EPI_RESID')[1] == PP                     # residual epilogue
    assert plan(T, 512, 512, 'TFX_EPI_F32')[1] == PP                       # fp32 outputs below K = 1024 (logits: K = 512)
    assert plan(T, 512, 1024, 'TFX_EPI_F32')[1] == OW                      # ... from K = 1024 on the one-wave kernel
    assert plan(T, 512, 512, a_rowmap=8)[1] == PP                          # row-gathered A: not through the one-wave kernels' buffer resources
    assert plan(T, 256, 512) == (0, GLDS, 512 * 2)             # one 256-column tile per row block: 256 tiles < 512 -> 128 x 128 tiles
    assert plan(8192, 384, 1024) == (0, MID, 64 * 3)           # latent projection: <= one 128 x 128 tile per CU
    assert plan(2048, 2048, 24576) == (0, MID, 256)            # AdaLN table backward
    assert plan(640, 3080, 1024) == (0, SKINNY, 10 * 25)       # mixed decode step: 490 tiles of 64 x 64 do not fit one round
    assert plan(640, 1024, 1024) == (0, DECODE, 10 * 16)       # ... 160 do
    assert plan(64, 5632, 1024) == (0, DECODE, 88)
    assert plan(64, 5632, 128)[1] == SKINNY                    # K < 8 x 32: no K split
    assert plan(T, 1546, 512)[1] == FALLBACK                   # N % 4 != 0
    assert plan(T, 512, 500)[0] == -1                          # K % 64 != 0 is refused


def test_generated_attention_loops_match_their_generator(tmp_path):
    """round 6: the tile loops of the attention kernels that run as generated asm (tools/gen_attn_loops.py: forward unmasked tiles, backward dQ whole loop; soft-cap
    plan modes 0 / 1) are committed under csrc/.  They must be what the generator writes today; MFMA counts per file (forward: 6 tiles x 16 with the fill and the
    first unit's missing P.V; dQ: 8 tiles x 24 + fill 8 + drain 4 - 4, + 4 in each of 15 phases for the dQ a stopping wave still owes); the LDS wait tracker never asks for more than the 4-bit counter holds; every fixed register
    the loops name is in the clobber list the kernels hand to hipcc."""
    import re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'transfusion_pytorch_amd', 'csrc')
    subprocess.run([sys.executable, os.path.join(root, 'tools', 'gen_attn_loops.py'), '--outdir', str(tmp_path)], check=True, capture_output=True)
    made = sorted(os.listdir(tmp_path))
    assert made == ['attn_asm_clobbers.inc', 'attn_dq_loop_m0.inc', 'attn_dq_loop_m1.inc', 'attn_fwd_loop_m0.inc', 'attn_fwd_loop_m1.inc']
    clob = set(re.findall(r'"v(\d+)"', open(os.path.join(csrc, 'attn_asm_clobbers.inc')).read()))
    for name in made:
        new, old = open(tmp_path / name).read(), open(os.path.join(csrc, name)).read()
        assert new == old, f'{name}: committed file differs from its generator\'s output'
        if name == 'attn_asm_clobbers.inc':
            continue
        lines = [ln.strip().strip('"').replace('\\n\\t', '') for ln in old.splitlines() if ln.startswith('"')]
        assert sum(ln.startswith('v_mfma') for ln in lines) == (260 if '_dq_' in name else 96), name     # dQ: + 4 per phase for a wave's first dead phase
        assert all(int(m) <= 15 for ln in lines for m in re.findall(r'lgkmcnt\((\d+)\)', ln)), name
        assert sum(ln == 's_barrier' for ln in lines) == (8 if '_dq_' in name else 6), name      # one barrier per tile, in every copy of the unrolled ring
        if '_dq_' in name:                                                                       # (the forward's fixed registers are bound / clobbered by hand in attention.hip)
            used = set(re.findall(r'\bv(\d+)\b', ' '.join(lines))) | {str(r) for a, b in re.findall(r'v\[(\d+):(\d+)\]', ' '.join(lines)) for r in range(int(a), int(b) + 1)}
            assert used <= clob, sorted(used - clob, key=int)


def test_generated_asm_loops_match_their_generators(tmp_path):
    """round 5: the K loops of the one-wave-per-SIMD GEMM kernels are generated inline-asm text (tools/gen_nt_ow_loop.py, tools/gen_tn_ow_loop.py), committed under
    csrc/ so that a build needs no generator run.  The committed files must be what the generators write today (an edit to one without the other would ship a
    stale schedule), every body must hold its 64 MFMAs per K-tile, and the loop's back branch must be the LAST instruction of its body (the one bug of the
    session: a bias-gradient MFMA emitted behind the branch)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'transfusion_pytorch_amd', 'csrc')
    for gen, first in (('gen_nt_ow_loop.py', 'gemm_nt_ow_loop.inc'), ('gen_tn_ow_loop.py', 'gemm_tn_ow_loop.inc')):
        subprocess.run([sys.executable, os.path.join(root, 'tools', gen), '--out', str(tmp_path / first)], check=True, capture_output=True)
    made = sorted(os.listdir(tmp_path))
    assert made == ['gemm_nt_ow_loop.inc', 'gemm_nt_owp_last.inc', 'gemm_nt_owp_next.inc', 'gemm_nt_owp_pro.inc', 'gemm_tn_ow_loop.inc', 'gemm_tn_ow_sum01.inc', 'gemm_tn_ow_sum23.inc']
    for name in made:
        new, old = open(tmp_path / name).read(), open(os.path.join(csrc, name)).read()
        assert new == old, f'{name}: committed file differs from its generator\'s output'
        lines = [ln.strip().strip('"').replace('\\n\\t', '') for ln in old.splitlines() if ln.startswith('"')]
        n_mfma = sum(ln.startswith('v_mfma') for ln in lines)
        if name != 'gemm_nt_owp_pro.inc':
            per_body = 72 if name.startswith('gemm_tn_ow_sum') else 64
            assert n_mfma % per_body == 0 and n_mfma >= 3 * per_body, (name, n_mfma)
        for i, ln in enumerate(lines):
            if ln.startswith('s_cbranch_scc1') and ('_loop_' in ln or '_steady_' in ln):      # a back branch: nothing of the body may follow it
                nxt = lines[i + 1]
                assert nxt.endswith(':') , (name, ln, nxt)


def test_tn_grouped_launch_plan():
    """`group_next` chains (host logic, no device): products over the same M rows that the one-wave kernel takes run as ONE launch - plan kind 3 with the chain's
    summed tiles and the row chunks that fill the chip for THAT tile count; a member it does not take (gathered rows), different M or a chain of five fall back to the
    head's own plan (the library then runs the chain product by product)."""
    lib = capi.lib()

    def args(M, N, K, **kw):
        return capi.make_args('tfx_gemm_tn_args', M=M, N=N, K=K, lda=(N + 7) // 8 * 8, a_cols=(N + 7) // 8 * 8, ldb=(K + 7) // 8 * 8, b_cols=(K + 7) // 8 * 8,
                              ldc=K, k_valid=K, splits=0, accumulate=1, alpha=1.0, **kw)

    def plan(chain):
        for a, b in zip(chain, chain[1:]):
            a.group_next = ctypes.addressof(b)
        out = [ctypes.c_int32(-9) for _ in range(4)]
        assert lib.tfx_gemm_tn_plan(ctypes.byref(chain[0]), *[ctypes.byref(o) for o in out]) == 0
        return tuple(o.value for o in out)

    T = 65536
    assert plan([args(T, 512, 1408), args(T, 2816, 512)]) == (3, 34, 7, 240)            # the FeedForward pair: 12 + 22 tiles, 7 chunks instead of 20 and 11
    assert plan([args(T, 512, 512), args(T, 1544, 512)]) == (3, 18, 13, 240)            # to_out + to_qk/v/gates: the 512 x 512 product on 256 x 256 tiles
    assert plan([args(T, 512, 512), args(T, 1544, 512), args(T, 512, 512), args(T, 512, 512)]) == (3, 26, 9, 240)      # + the skip projection's two halves
    assert plan([args(T, 512, 512), args(T, 1544, 512, a_rowmap=64)]) == (0, 16, 16, 256)          # a gathered member: the head's own plan
    assert plan([args(T, 512, 512), args(T // 2, 1544, 512)])[0] == 0                              # different row counts
    assert plan([args(T, 512, 512) for _ in range(7)])[0] == 0                                     # more than six members


def test_asm_loop_schedule_invariants():
    """Static checks of the generated K-loop bodies (tools/gen_nt_ow_loop.py, tools/gen_tn_ow_loop.py) - what a GPU run would only show as a wrong number or a hang:
    a steady K-tile issues exactly the 16 DMA pieces of the next-but-one K-tile and reads every fragment of a K-tile once; its counted `vmcnt` wait names exactly the
    pieces issued ahead of it (so the PREVIOUS K-tile's sixteen are what it certifies); every wave passes the same number of barriers in every loop form of a kernel
    (the bias-gradient forms share their blocks with the plain form); the scalar M0 / offset updates precede the DMA piece that uses them."""
    import importlib.util, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, 'tools', name + '.py'))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        return m

    def check_steady(lines, n_reads, n_mfma):
        assert sum(ln.startswith('v_mfma') for ln in lines) == n_mfma
        dma = [i for i, ln in enumerate(lines) if ln.startswith('buffer_load_dwordx4') and ln.endswith('lds')]
        assert len(dma) == 16
        assert sum(ln.startswith('ds_read') for ln in lines) == n_reads
        waits = [(i, ln) for i, ln in enumerate(lines) if ln.startswith('s_waitcnt vmcnt')]
        assert len(waits) == 1
        i, ln = waits[0]
        assert int(ln.split('(')[1].rstrip(')')) == sum(d < i for d in dma), ln
        nxt_read = next(k for k in range(i, len(lines)) if lines[k].startswith('ds_read'))
        assert 's_barrier' in lines[i:nxt_read]                                             # the wait is published by a barrier before anybody reads the K-tile it certifies
        for d in dma:                                                                        # m0 is set / advanced between two consecutive pieces
            prev = max([x for x in dma if x < d], default=-1)
            assert any('m0' in lines[k] and lines[k].startswith('s_') for k in range(prev + 1, d)), lines[d]
        return sum(ln == 's_barrier' for ln in lines)

    nt = load('gen_nt_ow_loop')
    bars = check_steady(nt.body('steady', loop=None), 32, 64)
    assert bars == 3
    assert check_steady(nt.body('steady', zero=True, loop=None), 32, 64) == 3
    assert sum(ln.startswith('ds_read') for ln in nt.body('t2')) == 16 and sum(ln == 's_barrier' for ln in nt.body('t2')) == 0
    tn = load('gen_tn_ow_loop')
    tn.SUM = False
    plain = check_steady(tn.body('steady'), 64, 64)
    tn.SUM, tn.SUM_BLOCKS, tn.SUMS, tn.ONES = True, (0, 1), 30, 38
    summed = check_steady(tn.body('steady'), 64, 72)
    assert plain == summed == 3
    for kind in ('t1', 't2'):
        tn.SUM = False; a = sum(ln == 's_barrier' for ln in tn.body(kind))
        tn.SUM = True; b = sum(ln == 's_barrier' for ln in tn.body(kind))
        assert a == b
    # a back branch is the LAST instruction of its body
    body = tn.body('steady', loop='L_x')
    assert body[-1] == 's_cbranch_scc1 L_x' and body[-2].startswith('s_cmp')
