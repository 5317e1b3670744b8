// Launch-list replay (include/tfx.h "launch lists"): the training / decode step is a static list of kernel launches over persistent
// buffers, so the host hands the whole list to the library once per step instead of paying one FFI round trip per kernel.
// Host code only; every case forwards to the public entry point of the same name.
#include <atomic>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include "../../include/tfx.h"

namespace {

inline int run_one(const tfx_launch& l, void* s) {
  const void* a = l.args;
  const tfx_raw_args* r = static_cast<const tfx_raw_args*>(a);
  switch (l.op) {
    case TFX_OP_GEMM_NT:          return tfx_gemm_nt(static_cast<const tfx_gemm_nt_args*>(a), s);
    case TFX_OP_GEMM_TN:          return tfx_gemm_tn(static_cast<const tfx_gemm_tn_args*>(a), s);
    case TFX_OP_ATTN_FWD:         return tfx_attn_fwd(static_cast<const tfx_attn_args*>(a), s);
    case TFX_OP_ATTN_BWD:         return tfx_attn_bwd(static_cast<const tfx_attn_args*>(a), s);
    case TFX_OP_ADALN_PRE_FWD:    return tfx_adaln_pre_fwd(static_cast<const tfx_adaln_pre_args*>(a), s);
    case TFX_OP_ADALN_PRE_BWD:    return tfx_adaln_pre_bwd(static_cast<const tfx_adaln_pre_args*>(a), s);
    case TFX_OP_ADALN_POST_FWD:   return tfx_adaln_post_fwd(static_cast<const tfx_adaln_post_args*>(a), s);
    case TFX_OP_ADALN_POST_BWD:   return tfx_adaln_post_bwd(static_cast<const tfx_adaln_post_args*>(a), s);
    case TFX_OP_QK_NORM_ROPE_FWD: return tfx_qk_norm_rope_fwd(static_cast<const tfx_qk_norm_rope_args*>(a), s);
    case TFX_OP_QK_NORM_ROPE_BWD: return tfx_qk_norm_rope_bwd(static_cast<const tfx_qk_norm_rope_args*>(a), s);
    case TFX_OP_ATTNRES_FWD:      return tfx_attnres_fwd(static_cast<const tfx_attnres_args*>(a), s);
    case TFX_OP_ATTNRES_BWD:      return tfx_attnres_bwd(static_cast<const tfx_attnres_args*>(a), s);
    case TFX_OP_RMSNORM_FWD:      return tfx_rmsnorm_fwd(static_cast<const tfx_rmsnorm_args*>(a), s);
    case TFX_OP_RMSNORM_BWD:      return tfx_rmsnorm_bwd(static_cast<const tfx_rmsnorm_args*>(a), s);
    case TFX_OP_EMBED_FWD:        return tfx_embed_fwd(static_cast<const tfx_embed_args*>(a), s);
    case TFX_OP_EMBED_BWD:        return tfx_embed_bwd(static_cast<const tfx_embed_args*>(a), s);
    case TFX_OP_NOISE_MIX:        return tfx_noise_mix(static_cast<const tfx_noise_mix_args*>(a), s);
    case TFX_OP_FOURIER:          return tfx_fourier(static_cast<const tfx_fourier_args*>(a), s);
    case TFX_OP_CE_FWD_BWD:       return tfx_ce_fwd_bwd(static_cast<const tfx_ce_args*>(a), s);
    case TFX_OP_MSE_FWD_BWD:      return tfx_mse_fwd_bwd(static_cast<const tfx_mse_args*>(a), s);
    case TFX_OP_CAST_ROWS:        return tfx_cast_rows(static_cast<const tfx_cast_args*>(a), s);
    case TFX_OP_CAST_ROWS_T:      return tfx_cast_rows_t(static_cast<const tfx_cast_args*>(a), s);
    case TFX_OP_ADAM_STEP:        return tfx_adam_step(static_cast<const tfx_adam_args*>(a), s);
    case TFX_OP_DECODE_ATTN:      return tfx_decode_attn(static_cast<const tfx_attn_args*>(a), s);
    // positional entry points: pointers p0.., integers i0.., floats f0 in declaration order
    case TFX_OP_OUTPUT_TO_FLOW:
      return tfx_output_to_flow((float*)r->p0, (const float*)r->p1, (const float*)r->p2, (const int32_t*)r->p3, (const float*)r->p4,
                                (int32_t)r->i0, (int32_t)r->i1, r->f0, s);
    case TFX_OP_GATHER_F32:
      return tfx_gather_f32((const float*)r->p0, (const int32_t*)r->p1, (float*)r->p2, (int32_t)r->i0, s);
    case TFX_OP_ONEHOT_BF16:
      return tfx_onehot_bf16((const int32_t*)r->p0, (const int32_t*)r->p1, (tfx_bf16*)r->p2, (int32_t)r->i0, (int32_t)r->i1, s);
    case TFX_OP_SCATTER_ROWS_BF16:
      return tfx_scatter_rows_bf16((const tfx_bf16*)r->p0, (int32_t)r->i0, (int32_t)r->i1, (tfx_bf16*)r->p1, (int32_t)r->i2,
                                   (const int32_t*)r->p2, (int32_t)r->i3, s);
    case TFX_OP_F32_TO_BF16:
      return tfx_f32_to_bf16((const float*)r->p0, (tfx_bf16*)r->p1, r->i0, s);
    case TFX_OP_SILU_BWD:
      return tfx_silu_bwd((const tfx_bf16*)r->p0, (const tfx_bf16*)r->p1, (tfx_bf16*)r->p2, r->i0, s);
    case TFX_OP_COLSUM_BF16:
      return tfx_colsum_bf16((const tfx_bf16*)r->p0, (int32_t)r->i0, (int32_t)r->i1, (int32_t)r->i2, (const int32_t*)r->p1,
                             (const int32_t*)r->p2, (float*)r->p3, s);
    case TFX_OP_COLSUM_F32:
      return tfx_colsum_f32((const float*)r->p0, (int32_t)r->i0, (int32_t)r->i1, (int32_t)r->i2, (float*)r->p1, s);
    case TFX_OP_ADD_BF16:
      return tfx_add_bf16((const tfx_bf16*)r->p0, (const tfx_bf16*)r->p1, (tfx_bf16*)r->p2, r->i0, s);
    case TFX_OP_ADALN_POST_PRE_FWD:
      return tfx_adaln_post_pre_fwd((const tfx_adaln_post_args*)r->p0, (const tfx_adaln_pre_args*)r->p1, s);
    case TFX_OP_LAYER_END_FWD:
      return tfx_layer_end_fwd((const tfx_adaln_post_args*)r->p0, (const tfx_attnres_args*)r->p1, (const tfx_adaln_pre_args*)r->p2, s);
    case TFX_OP_ATTNRES_PREP:
      return tfx_attnres_prep((const tfx_attnres_src*)r->p0, (int32_t)r->i0, (int32_t)r->i1, s);
    case TFX_OP_ATTNRES_FINISH:
      return tfx_attnres_finish((const tfx_attnres_src*)r->p0, (int32_t)r->i0, (int32_t)r->i1, s);
    case TFX_OP_ATTNRES_PULL_BWD:
      return tfx_attnres_pull_bwd((const tfx_attnres_pull_args*)r->p0, (const tfx_adaln_post_args*)r->p1, s);
    case TFX_OP_ADALN_PRE_POST_BWD:
      return tfx_adaln_pre_post_bwd((const tfx_adaln_pre_args*)r->p0, (const tfx_adaln_post_args*)r->p1, s);
    case TFX_OP_CAST_BLOCK_BF16:
      return tfx_cast_block_bf16((const float*)r->p0, (int32_t)r->i0, (tfx_bf16*)r->p1, (int32_t)r->i1, (int32_t)r->i2, (int32_t)r->i3, s);
    case TFX_OP_SCALE_BF16_COPY:
      return tfx_scale_bf16_copy((const tfx_bf16*)r->p0, (tfx_bf16*)r->p1, r->i0, r->f0, s);
    case TFX_OP_SCALE_BF16_DEV:
      return tfx_scale_bf16_dev((tfx_bf16*)r->p0, r->i0, (const float*)r->p1, s);
    default: return -100;          // unknown op
  }
}

// the side stream and its fork / join events: one set per process (one process drives one GPU), created on first use
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork_ev[64] = {}, join_ev[64] = {};
  int ensure() {
    if (stream) return 0;
    // non-blocking: no implicit sync with the null stream.  LOWEST priority: the weight-gradient GEMMs fill what the data-gradient chain
    // leaves idle instead of competing with it
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = least urgent (numerically greatest)
    if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, lo) != hipSuccess) return -110;
    for (int i = 0; i < 64; ++i)
      if (hipEventCreateWithFlags(&fork_ev[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming) != hipSuccess) return -111;
    return 0;
  }
};
// one side stream (and its event slots) PER DEVICE: a process that drives several GPUs - or two models on two devices - gets an independent
// set for each; the list is replayed on the device that is current when tfx_run_list is called (the caller's stream lives there)
constexpr int kMaxDevices = 16;
SideStream g_sides[kMaxDevices];
// the switch of tfx_set_single_stream is PROCESS-wide on purpose: PyTorch runs `loss.backward()` on its autograd worker thread, so a caller that
// sets it on the main thread (bench.py's bracketed steps, the one-stream parity test) must reach the replay of the backward list on another thread (a
// thread_local switch, tried first in round 5, silently left the weight-gradient GEMMs on the side stream there).  What must NOT leak to other threads
// is the temporary single-stream mode of a graph capture (engine.replay_auto used to flip the global around tfx_graph_create): that one is
// the thread-local override below, set only inside tfx_graph_create_single.
std::atomic<bool> g_single_stream{false};
thread_local int t_force_single = 0;
inline bool single_stream() { return t_force_single > 0 || g_single_stream.load(std::memory_order_relaxed); }
inline SideStream* side_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return &g_sides[dev];
}

}  // namespace

extern "C" int tfx_run_list(const tfx_launch* list, int32_t n, void* stream, int32_t* failed_at) {
  if (n < 0 || (n > 0 && !list)) return -1;
  hipStream_t main_s = (hipStream_t)stream;
  // resolved on first use: a list without side work (and the empty list) never asks the runtime which device is current
  SideStream* sp = nullptr;
  auto side = [&](int& rc) -> SideStream* {
    if (!sp && !(sp = side_of_current_device())) { rc = -112; return nullptr; }
    rc = sp->ensure();
    return rc == 0 ? sp : nullptr;
  };
  for (int32_t i = 0; i < n; ++i) {
    const tfx_launch& l = list[i];
    int rc = 0;
    if (!l.args) rc = -2;
    else if (l.op >= TFX_OP_FORK && l.op <= TFX_OP_JOIN_WAIT) {
      if (l.stream < 0 || l.stream >= 64) rc = -3;
      else if (single_stream()) rc = 0;
      else if (SideStream* gs = side(rc)) {
        SideStream& g_side = *gs;
        if (l.op == TFX_OP_FORK) {
          rc = (int)hipEventRecord(g_side.fork_ev[l.stream], main_s);
          if (rc == 0) rc = (int)hipStreamWaitEvent(g_side.stream, g_side.fork_ev[l.stream], 0);
        } else {
          if (l.op != TFX_OP_JOIN_WAIT) rc = (int)hipEventRecord(g_side.join_ev[l.stream], g_side.stream);
          if (rc == 0 && l.op != TFX_OP_JOIN_RECORD) rc = (int)hipStreamWaitEvent(main_s, g_side.join_ev[l.stream], 0);
        }
      }
    } else if (l.stream == 1 && !single_stream()) {
      if (SideStream* gs = side(rc)) rc = run_one(l, (void*)gs->stream);
    } else if (l.stream == 1) {
      rc = run_one(l, stream);
    } else if (l.stream == 0) {
      rc = run_one(l, stream);
    } else rc = -3;
    if (rc != 0) { if (failed_at) *failed_at = i; return rc; }
  }
  return 0;
}

// ---- hipGraph form of a launch list (decode plans) ------------------------------------------------------------------------------
namespace {
hipStream_t g_capture_streams[kMaxDevices] = {};
}
extern "C" int tfx_graph_create(const tfx_launch* list, int32_t n, void** graph_out) {
  if (!graph_out || n <= 0 || !list) return -1;
  *graph_out = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -112;
  hipStream_t& g_capture_stream = g_capture_streams[dev];
  if (!g_capture_stream && hipStreamCreateWithFlags(&g_capture_stream, hipStreamNonBlocking) != hipSuccess) return -120;
  // thread-local mode: other host threads (and other streams of this thread) keep working while the list is recorded
  if (hipStreamBeginCapture(g_capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -121;
  int32_t failed = -1;
  const int rc = tfx_run_list(list, n, (void*)g_capture_stream, &failed);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(g_capture_stream, &graph);
  if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess || !graph) return -122;
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) return -123;
  *graph_out = (void*)exec;
  return 0;
}
extern "C" int tfx_graph_create_single(const tfx_launch* list, int32_t n, void** graph_out) {
  ++t_force_single;                      // this thread's capture only: other threads keep replaying / fingerprinting in their own mode
  const int rc = tfx_graph_create(list, n, graph_out);
  --t_force_single;
  return rc;
}
extern "C" int tfx_graph_launch(void* graph, void* stream) {
  if (!graph) return -1;
  return (int)hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream);
}
extern "C" int tfx_graph_destroy(void* graph) {
  if (!graph) return 0;
  return (int)hipGraphExecDestroy((hipGraphExec_t)graph);
}

// ---- fingerprint of a launch list: has anything a capture would freeze changed since the last replay? ----------------------------------
namespace {
inline size_t struct_bytes(int op) {
  switch (op) {
    case TFX_OP_GEMM_NT: return sizeof(tfx_gemm_nt_args);
    case TFX_OP_GEMM_TN: return sizeof(tfx_gemm_tn_args);
    case TFX_OP_ATTN_FWD: case TFX_OP_ATTN_BWD: case TFX_OP_DECODE_ATTN: return sizeof(tfx_attn_args);
    case TFX_OP_ADALN_PRE_FWD: case TFX_OP_ADALN_PRE_BWD: return sizeof(tfx_adaln_pre_args);
    case TFX_OP_ADALN_POST_FWD: case TFX_OP_ADALN_POST_BWD: return sizeof(tfx_adaln_post_args);
    case TFX_OP_QK_NORM_ROPE_FWD: case TFX_OP_QK_NORM_ROPE_BWD: return sizeof(tfx_qk_norm_rope_args);
    case TFX_OP_ATTNRES_FWD: case TFX_OP_ATTNRES_BWD: return sizeof(tfx_attnres_args);
    case TFX_OP_RMSNORM_FWD: case TFX_OP_RMSNORM_BWD: return sizeof(tfx_rmsnorm_args);
    case TFX_OP_EMBED_FWD: case TFX_OP_EMBED_BWD: return sizeof(tfx_embed_args);
    case TFX_OP_NOISE_MIX: return sizeof(tfx_noise_mix_args);
    case TFX_OP_FOURIER: return sizeof(tfx_fourier_args);
    case TFX_OP_CE_FWD_BWD: return sizeof(tfx_ce_args);
    case TFX_OP_MSE_FWD_BWD: return sizeof(tfx_mse_args);
    case TFX_OP_CAST_ROWS: case TFX_OP_CAST_ROWS_T: return sizeof(tfx_cast_args);
    case TFX_OP_ADAM_STEP: return sizeof(tfx_adam_args);
    default: return op >= TFX_OP_OUTPUT_TO_FLOW ? sizeof(tfx_raw_args) : 0;
  }
}
inline uint64_t mix(uint64_t h, const void* p, size_t n) {           // FNV-1a over 8-byte words (+ tail bytes)
  const unsigned char* b = static_cast<const unsigned char*>(p);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t w; __builtin_memcpy(&w, b + i, 8); h = (h ^ w) * 1099511628211ull; }
  for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
}  // namespace
extern "C" int tfx_list_fingerprint(const tfx_launch* list, int32_t n, int64_t* out) {
  if (!out || n < 0 || (n > 0 && !list)) return -1;
  uint64_t h = 1469598103934665603ull ^ (g_single_stream.load(std::memory_order_relaxed) ? 0x9e3779b97f4a7c15ull : 0ull);
  for (int32_t i = 0; i < n; ++i) {
    const tfx_launch& l = list[i];
    h = mix(h, &l.op, 4); h = mix(h, &l.stream, 4);
    if (l.op >= TFX_OP_FORK && l.op <= TFX_OP_JOIN_WAIT) continue;
    if (!l.args) return -2;
    const size_t nb = struct_bytes(l.op);
    if (nb == 0) return -100;
    h = mix(h, l.args, nb);
    // entry points that take pointers to HOST structs: what they point at is part of the frozen state
    const tfx_raw_args* r = static_cast<const tfx_raw_args*>(l.args);
    switch (l.op) {
      case TFX_OP_ADALN_POST_PRE_FWD: h = mix(h, r->p0, sizeof(tfx_adaln_post_args)); h = mix(h, r->p1, sizeof(tfx_adaln_pre_args)); break;
      case TFX_OP_LAYER_END_FWD:
        h = mix(h, r->p0, sizeof(tfx_adaln_post_args)); h = mix(h, r->p1, sizeof(tfx_attnres_args));
        if (r->p2) h = mix(h, r->p2, sizeof(tfx_adaln_pre_args));
        break;
      case TFX_OP_ATTNRES_PULL_BWD: h = mix(h, r->p0, sizeof(tfx_attnres_pull_args)); if (r->p1) h = mix(h, r->p1, sizeof(tfx_adaln_post_args)); break;
      case TFX_OP_ADALN_PRE_POST_BWD: h = mix(h, r->p0, sizeof(tfx_adaln_pre_args)); h = mix(h, r->p1, sizeof(tfx_adaln_post_args)); break;
      case TFX_OP_GEMM_TN: {                                        // a `group_next` chain: the chained structs are copied by value into the launch a capture freezes
        const tfx_gemm_tn_args* g = static_cast<const tfx_gemm_tn_args*>(l.args);
        for (int k = 0; k < 8 && g->group_next; ++k) { g = static_cast<const tfx_gemm_tn_args*>(g->group_next); h = mix(h, g, sizeof(tfx_gemm_tn_args)); }
        break;
      }
      default: break;
    }
  }
  *out = (int64_t)h;
  return 0;
}

extern "C" int tfx_set_single_stream(int32_t on) {
  const int prev = g_single_stream.exchange(on != 0) ? 1 : 0;
  return prev;
}
