"""The packing seam of the reference (`transfusion_pytorch/modality_processing.py`): a registry of callables

    PROCESSING_STRATEGIES[name](modalities, times, model, *, need_axial_pos_emb, return_loss, return_embed) -> ProcessedModalityBatch

(MP:138-147 the result type, MP:1050-1058 the registry, MP:1254-1256 the lookup, called at T:3107-3126).  The reference ships four
strategies plus a timing-based `auto` router that all produce the same packed batch (its own equivalence test: MP:1258-1305); here every
name maps to the ONE native packer: the host structure scan (packing.scan_batch, MP:206-377) plus two HIP launches per modality type -
`tfx_noise_mix` (x_t = t x + (1 - t) eps, flow = x - eps, MP:654-656) and the row-scattered `latent_to_model` GEMM (T:1478); axial positional
embeddings are evaluated per instance by the model's MLP, `pre_post_transformer_enc_dec` types go through the user's encoder (projected lengths, MP:738-741).

`Transfusion.forward` does not go through this function (its plan fuses the same launches into the step's launch list); it is the drop-in
entry point for callers that use the registry directly, and the tests compare the two.  `forward` does CONSULT the registry: if the entry of the
model's strategy name is no longer the native packer it raises instead of ignoring the replacement (tests/test_decode_contract_gpu.py).
"""
from __future__ import annotations

import ctypes
from typing import Callable, NamedTuple

import numpy as np
import torch
import torch.nn.functional as F

from . import capi
from .packing import scan_batch
from .params import pad_to


class ProcessedModalityBatch(NamedTuple):                     # MP:138-147
    text: torch.Tensor                                        # Int['b n']: token ids, -1 on latent slots and padding
    modality_tokens: torch.Tensor                             # Float['b n d']: projected (noised) latents at their slots, zeros elsewhere
    modality_positions: list                                  # per sample [(type, offset, length)]
    modality_pos_emb: list | None
    flows: dict                                               # type -> [flow target (x - eps) per instance]   (return_loss only)
    get_pred_flows: list                                      # type -> [fn(embed, need_splice=True) -> (*axial, d)]   MP:160-175
    get_recon_losses: dict                                    # type -> [fn(pred_flow) -> scalar]   MP:177-200
    pos_emb_max_axial_dims: dict
    total_tokens: int | None


def process_native(modalities, times, model, *, need_axial_pos_emb: bool = False, return_loss: bool = True, return_embed: bool = False,
                   noise: dict | None = None) -> ProcessedModalityBatch:
    """modalities: list of samples as `forward` hands them to the packer - [sos] / [eos] already added when training (T:3016-3023);
    times: Float['b m'].  `noise` (type -> (R, dim_latent), scan order) replaces the `randn_like` draw (MP:654) - the parity tests inject it."""
    model._require_gpu()
    md, dev, stream = model.md, model.device, model._stream()
    sp = ctypes.c_void_p(stream)
    lib = capi.lib()
    ext_ctx = None
    if model._ext:        # `pre_post_transformer_enc_dec` types: noising + the user's encoder in PyTorch, shape placeholders of the PROJECTED shape for the scan (MP:715-745)
        modalities, ext_ctx = model._ext_preprocess(modalities, times.to(dev, torch.float32), return_loss)
    P = scan_batch(modalities, num_modalities=model.num_modalities, dim_latents=model.dim_latents, sos_id=model.sos_id, eos_id=model.eos_id,
                   meta_id=model.meta_id, som_ids=model.som_ids, eom_ids=model.eom_ids, add_sos_eos=False, add_meta=not return_embed)
    b, n, d = P.b, P.n_full, md.dim
    text = torch.from_numpy(P.text_host.astype(np.int64)).to(dev)
    if P.user_text:
        text.view(-1).index_copy_(0, torch.from_numpy(P.text_dest).to(dev), torch.cat([t.reshape(-1) for t in P.user_text]).to(dev, torch.int64))
    tokens = torch.zeros(b * n, d, device=dev, dtype=torch.bfloat16)
    model.store.refresh_shadows(stream)
    inst_time = times.to(dev, torch.float32)[torch.from_numpy(P.inst_b).to(dev), torch.from_numpy(P.inst_m).to(dev)] if len(P.inst_b) else \
        torch.zeros(1, device=dev)
    flows, recon = {}, {}
    per_type = {}
    # (the dict is filled below; `None` marks a type whose rows come from the user's encoder)
    for t, lat in P.latents.items():
        dl, dlp = md.dim_latents[t], pad_to(md.dim_latents[t], 64)
        if t in model._ext:                                   # rows straight from the user's encoder
            rows = torch.cat(ext_ctx[t]['tok']).detach().to(torch.bfloat16).contiguous()
            row_pos = torch.from_numpy(P.row_pos[t]).to(dev)
            capi.check(lib.tfx_scatter_rows_bf16(rows.data_ptr(), d, d, tokens.data_ptr(), d, row_pos.data_ptr(), rows.shape[0], sp), 'tfx_scatter_rows_bf16')
            per_type[t] = None
            continue
        x = torch.cat(lat).to(dev, torch.float32).contiguous()
        R = x.shape[0]
        eps = (noise[t].to(dev, torch.float32).contiguous() if noise is not None else torch.randn_like(x)) if return_loss else None
        xt = torch.zeros(R, dlp, device=dev, dtype=torch.bfloat16)
        flow = torch.empty(R, dl, device=dev, dtype=torch.float32) if return_loss else None
        row_inst = torch.from_numpy(P.row_inst[t]).to(dev)
        row_pos = torch.from_numpy(P.row_pos[t]).to(dev)
        a = capi.make_args('tfx_noise_mix_args', R=R, dl=dl, x=x, eps=capi.ptr(eps), row_inst=row_inst, inst_time=inst_time, xt=xt, ld_xt=dlp, flow=capi.ptr(flow))
        capi.call('tfx_noise_mix', a, stream)
        if dl == d:                                           # nn.Identity latent_to_model (T:1478)
            capi.check(lib.tfx_scatter_rows_bf16(xt.data_ptr(), dlp, d, tokens.data_ptr(), d, row_pos.data_ptr(), R, sp), 'tfx_scatter_rows_bf16')
        else:
            g = capi.make_args('tfx_gemm_nt_args', A=xt, lda=dlp, B=model.store.shadows[f'in{t}'], ldb=dlp, M=R, N=d, K=dlp, epi=capi.ENUMS['TFX_EPI_BF16'],
                               C=tokens, ldc=d, bias=model.store.ptr(f'latent_to_model_projs.{t}.bias'), rowmap=row_pos)
            capi.call('tfx_gemm_nt', g, stream)
        per_type[t] = (x, eps, flow)
    # closures, in scan order per type (build_record_closures, MP:764-805)
    get_pred_flows = model._pred_flow_closures(P)
    cursor = {t: 0 for t in P.latents}
    for gi in range(len(P.inst_b)):
        t, L, shape = int(P.inst_type[gi]), int(P.inst_len[gi]), tuple(P.inst_shape[gi])
        if per_type[t] is None:                               # user-encoder type: targets in the raw layout, kept by the pre-processing
            if return_loss:
                k = len(flows.setdefault(t, []))
                c = ext_ctx[t]
                flows[t].append(c['flow'][k])
                recon.setdefault(t, []).append(lambda pred_flow, nz=c['noised'][k], eps=c['eps'][k], tt=c['time'][k]: F.mse_loss(nz, eps + pred_flow * (1. - tt)))
            continue
        x, eps, flow = per_type[t]
        lo = cursor[t]; cursor[t] += L
        if return_loss:
            dl = md.dim_latents[t]
            tt = inst_time[gi]
            flows.setdefault(t, []).append(flow[lo:lo + L].view(*shape, dl))

            def recon_fn(pred_flow, x=x, eps=eps, lo=lo, L=L, shape=shape, dl=dl, tt=tt):      # get_recon_loss, MP:177-200
                noised = (x[lo:lo + L] * tt + eps[lo:lo + L] * (1. - tt)).view(*shape, dl)
                return F.mse_loss(noised, eps[lo:lo + L].view(*shape, dl) + pred_flow * (1. - tt))
            recon.setdefault(t, []).append(recon_fn)
    pos_emb, max_dims = None, {}
    if need_axial_pos_emb:
        # the reference hands back a lazy description + the per-type maximum extents and evaluates it right after (MP:1003-1045); here the rows are
        # evaluated at once: (b, n, d), the embedding of an instance at its slots, zeros on text / meta tokens
        pos = torch.zeros(b * n, d, device=dev)
        for t in P.latents:
            if model.pos_emb_mlp[t] is None:
                continue
            shapes = [P.inst_shape[g] for g in range(len(P.inst_b)) if int(P.inst_type[g]) == t]
            pos.index_copy_(0, torch.from_numpy(P.row_pos[t].astype(np.int64)).to(dev), model._pos_rows(t, shapes).float())
            max_dims[t] = [torch.tensor(sh) for sh in shapes]
        pos_emb = pos.view(b, n, d)
    return ProcessedModalityBatch(text=text, modality_tokens=tokens.view(b, n, d).float(), modality_positions=P.positions, modality_pos_emb=pos_emb,
                                  flows=flows, get_pred_flows=get_pred_flows, get_recon_losses=recon, pos_emb_max_axial_dims=max_dims,
                                  total_tokens=int(P.total_tokens))


# every reference strategy name resolves to the native packer (MP:1050-1058); `auto` needs no timing router: there is one implementation
PROCESSING_STRATEGIES: dict[str, Callable[..., ProcessedModalityBatch]] = {name: process_native for name in ('naive', 'grouped', 'flat', 'hybrid', 'auto')}


def process_modalities(modalities, times, model, *, need_axial_pos_emb=False, return_loss=True, return_embed=False) -> ProcessedModalityBatch:
    """the reference's dispatch helper (MP:1222-1256): look the model's strategy up in the registry and call it"""
    name = model.modality_processing
    assert name in PROCESSING_STRATEGIES, f'unknown modality processing strategy `{name}`, available: {list(PROCESSING_STRATEGIES)}'
    return PROCESSING_STRATEGIES[name](modalities, times, model, need_axial_pos_emb=need_axial_pos_emb, return_loss=return_loss, return_embed=return_embed)
