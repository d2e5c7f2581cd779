// capi.hip -- the extern "C" boundary of libvo_hip.so (see include/vo_hip.h).
// Host-side plumbing only: argument checks, device buffers, uploads, launches.
// There is no CPU implementation of any operator behind these entry points.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>

#include "../../include/vo_hip.h"
#include "vo_internal.h"
#include "../../include/vo/epipolar.hpp"   // host linear algebra of vo_estimate_transform

using namespace vo;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
void set_error_v(const char* fmt, va_list ap) { vsnprintf(g_err, sizeof(g_err), fmt, ap); }

#define VO_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      (void)hipGetLastError(); /* reported here: must not resurface in the next call's launch check */ \
      return fail(_e == hipErrorOutOfMemory ? VO_ERR_OUT_OF_MEMORY : VO_ERR_HIP,        \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    }                                                                                   \
  } while (0)

#define VO_REQUIRE(cond, msg) \
  do { if (!(cond)) return fail(VO_ERR_INVALID_ARG, "%s: %s", __func__, msg); } while (0)

// device arrays move as 8-byte pieces (vo_hip.h, Conventions): true when every pointer given is null or on such a boundary
template <typename... P>
static bool aligned8(const P*... p) { return ((... | reinterpret_cast<uintptr_t>(p)) & 7u) == 0; }

// entry points that copy host memory or wait for the stream cannot be part of a graph capture (vo_ctx_begin_capture): refused
// BEFORE any HIP call, because a refused HIP call would invalidate the capture in progress
#define VO_NOT_CAPTURING(ctx) \
  do { if ((ctx)->capturing) return fail(VO_ERR_NOT_READY, "%s: not allowed inside a graph capture (only *_dev entry points are)", __func__); } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // grow-only; contents are NOT preserved
  hipError_t ensure(size_t bytes, hipStream_t st) {
    if (bytes <= cap) return hipSuccess;
    {   // growing frees, allocates and synchronises: none of it may happen inside a graph capture
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return hipErrorStreamCaptureUnsupported;
    }
    if (p) {
      hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) return e;
      (void)hipFree(p);
      p = nullptr; cap = 0;
    }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

int vo_fail(int code, const char* fmt, ...) {        // for the other translation units (vo_internal.h)
  va_list ap;
  va_start(ap, fmt);
  set_error_v(fmt, ap);
  va_end(ap);
  return code;
}

struct vo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 0;
  char name[128] = "";
  DevBuf scratch;     // compaction counts
  DevBuf best;        // matcher keys
  DevBuf table;       // join table
  DevBuf in[6];       // staging for host-pointer entry points
  DevBuf out[3];
  DevBuf counts;      // small device ints
  DevBuf batch_pack;  // packed correspondences of the batched solver
  DevBuf batch_bad;   // its per-problem bad-index counters
  DevBuf batch_states, batch_partials;   // launch-per-round form of the batched solver: per-problem state + partial rows
  DevBuf batch_help;  // shared form (fewer problems than CUs): tagged words handed between workgroups
  PicpParams batch_params_host{};        //   its parameter block as last uploaded, and where
  const PicpParams* batch_params_dev = nullptr;
  DevBuf prune_ws;    // sorted copies / tables of the matcher's sorted variants
  DevBuf epi_ws;      // maxima, A^T A and vote counts of vo_estimate_transform
  // Steering of the automatic matcher by what the previous batched call found: after a call that ran the
  // exact-duplicate pass, "did any frame take it?" travels to pinned host memory behind the call (no wait); a later call
  // that finds the answer "none" there leaves the pass out for the next HINT_SKIP calls, then probes again.  Data whose
  // descriptors are recomputed per frame (no bitwise copies, ever) then pays the sampling pass once in seventeen calls.
  // A call that ran WITHOUT the pass asks the same question of the keys the search left (a sample query at distance 0 from
  // its match): the answer "yes" ends the skipping at once, so data that regains its copies loses one or two calls, not sixteen.
  // vo_match_set_mode() forgets what was learnt.
  int* hint_host = nullptr;        // pinned
  int* hint_dev = nullptr;
  hipEvent_t hint_ev = nullptr;
  bool hint_pending = false;
  int hint_skip_left = 0;
  bool hint_of_skipped_call = false;   // what the pending answer was asked of
  int hint_none_streak = 0;            // consecutive answers "no frame took the pass": two in a row start the skipping (a context
                                       // that alternates between data with and without copies is then never steered wrong)
  int match_mode = 0; // 0 auto, 1 full scan, 2 bucket-pruned scan, 3 cell-hash search, 4 / 5 exact-duplicate pass first, then 2 / 3
  int batch_form = 0; // batched solver: 0 auto, 1 one launch per round, 2 one workgroup per problem
  int batch_last_form = 0, batch_last_wgs = 0;   // what the last batched call ran as (vo_picp_batch_info)
  bool capturing = false;
  unsigned long long id = 0;   // unique per context ever created: an address can be reused, an id cannot
};

struct vo_event {
  int device = 0;
  hipEvent_t ev = nullptr;
};

// Contexts that exist.  Solvers, graphs and kd-trees keep a pointer to the context they were made on; the header asks for
// them to be destroyed first, but destruction order is easy to get wrong in a host program (statics, members, scripting
// languages), so a handle that outlives its context must fail cleanly when used and must still be destroyable.
static std::mutex g_ctx_mu;
static std::set<const vo_ctx*> g_ctx_live;
static unsigned long long g_ctx_next_id = 1;
static bool ctx_alive(const vo_ctx* c) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  return c && g_ctx_live.count(c) != 0;
}
// the context a handle was made on: still there, and still THAT context (not a new one at the recycled address)
static bool ctx_alive(const vo_ctx* c, unsigned long long id) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  return c && g_ctx_live.count(c) != 0 && c->id == id;
}

struct vo_graph {
  vo_ctx* ctx = nullptr;
  unsigned long long ctx_id = 0;
  hipGraphExec_t exec = nullptr;
};

struct vo_picp {
  vo_ctx* ctx = nullptr;
  unsigned long long ctx_id = 0;
  PicpParams hp;               // host mirror of the device parameters
  PicpParams* d_params = nullptr;
  PicpState* d_state = nullptr;
  bool params_dirty = true;
  DevBuf world_own, meas_own, pairs_own, packed, partials;
  const float* d_world = nullptr;
  const float* d_meas = nullptr;
  int n_world = 0, n_meas = 0;
  bool have_points = false;
  // host copy of the correspondences already uploaded and packed: a call whose pairs compare equal to it
  // (whole array, memcmp) skips the upload; anything else -- including an in-place edit -- is re-uploaded
  std::vector<int32_t> shadow;
  bool shadow_valid = false;
  bool packed_valid = false;
  int exact = 0;              // reference-order arithmetic (picp_exact_kernel)
  int set_n = -1;             // pair count handed over by vo_picp_set_correspondences
  int grid = 1;
  const float* pending_T0 = nullptr;   // device 4x4 to load as the pose by the next pack launch
  int zeroed_for_grid = -1;   // grid the (zero-padded) partial buffers were last cleared for
  // Rounds enqueued since the state (pose[0], T16, H, b, statistics) was last complete: vo_picp_one_round enqueues ONE
  // launch per call and leaves the finishing launch to whoever needs the state next (a getter, a setter, a multi-round
  // solve): picp_flush.  0 = the state is complete.
  int chain_len = 0;
  // Rounds enqueued BEYOND chain_len: once a caller has shown the reference's pattern -- oneRound after oneRound on one
  // unchanged vector -- a call enqueues its own round and up to run_ahead more as ONE graph launch (a hipGraphLaunch of 1..8
  // kernels costs the host 5.5-5.9 us whatever their number, a plain launch 3.0: tools/micro/host_call_cost.hip), and the next
  // calls find their round already in the stream: they only compare.  Round a writes slot a % PICP_SLOTS of the partial rows
  // and of the pose.  The call that enqueues rounds k .. k + run_ahead may still find its OWN comparison differing and must then
  // repeat round k from slot (k - 1) % PICP_SLOTS: so 1 + run_ahead <= PICP_SLOTS - 1 -- the rounds it enqueued never write the
  // slot of the last counted round; later calls, a getter's finishing launch, or a repeat on other pairs / points / parameters
  // read the slot of a round that was claimed, which nothing enqueued after it has written.  Whatever ran ahead and is not
  // claimed by a matching call is overwritten or ignored.  At most run_ahead rounds of GPU time are spent for nothing when
  // the loop ends.
  int ahead = 0;
  int streak = 0;                // consecutive one_round calls that matched their speculation
  int loop_hint = 0;             // rounds the previous run of calls had when a getter closed it: the reference's loop has a fixed
                                 // count (vo_complete.cpp:163: 100), so the last window of a run is cut to what is left of it and
                                 // nothing runs ahead for nothing; a longer run than last time simply goes on in full windows
  int run_ahead = PICP_SLOTS / 2 - 1;   // VO_PICP_RUN_AHEAD (0: every call enqueues exactly its own round; at most PICP_SLOTS - 2).
                                        // Windows of PICP_SLOTS / 2 rounds start at two alternating slots: two graphs per geometry.
  // A window's graph is captured only the third time it is wanted (key -> times wanted): a caller whose correspondence count --
  // and with it the launch geometry -- changes from frame to frame never pays a capture (~0.15 ms each), it gets one plain
  // launch per call as before (apps/one_round_rate, with_init_varying_sizes: 54 k iterations/s with eager captures, 115-120 k
  // without); and nothing is ever destroyed to make room: a full cache means plain launches for new keys.
  std::map<std::tuple<int, int, const void*, size_t, const void*, int>, int> chain_wanted;
  unsigned long long spec_rounds = 0, spec_redone = 0;   // one_round calls enqueued before their comparison / found different
  int use_graph = 1;
  int graph_failures = 0;     // captures that failed (the handle then stays on plain launches): vo_picp_graph_info
  std::map<std::tuple<int, int, const void*, size_t, const void*, int>, hipGraphExec_t> graphs;
};

static int set_device(vo_ctx* ctx) {
  VO_HIP_CHECK(hipSetDevice(ctx->device));
  return VO_OK;
}

// the `it` argument of round a >= 1 of an open chain (and of the finishing launch after a rounds): only its residue modulo
// PICP_SLOTS matters to the kernels (which slot they read and write), so a chain of any length cycles through 1 .. PICP_SLOTS
static int chain_it(int a) { return a <= 0 ? 0 : ((a - 1) % PICP_SLOTS) + 1; }

// the finishing launch of the rounds vo_picp_one_round has enqueued so far (vo_picp::chain_len), if any
static int picp_flush(vo_picp* s) {
  if (s->chain_len == 0) { s->ahead = 0; return VO_OK; }
  vo_ctx* c = s->ctx;
  if (c->capturing)
    return fail(VO_ERR_NOT_READY, "vo_picp: rounds of vo_picp_one_round are still open (read the pose or the statistics once before the capture)");
  PackedCorr pk{s->packed.as<float>(), s->packed.cap / (5 * sizeof(float)) & ~(size_t)3};
  // (rounds that ran ahead of the caller sit in front of this launch in the stream; they wrote other slots than the one it reads)
  VO_HIP_CHECK(launch_picp_finish(c->stream, s->d_params, s->d_state, pk, s->partials.as<float>(), s->grid, chain_it(s->chain_len)));
  s->loop_hint = s->chain_len;
  s->chain_len = 0;
  s->ahead = 0;
  return VO_OK;
}

extern "C" {

int vo_abi_version(void) { return VO_HIP_ABI_VERSION; }
const char* vo_last_error(void) { return g_err; }

int vo_ctx_create(int device, void* stream, vo_ctx** out) {
  VO_REQUIRE(out != nullptr, "out is null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(VO_ERR_NO_DEVICE, "no HIP device available (%s); libvo_hip has no CPU fallback",
                e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(VO_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, n);
  VO_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  VO_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(VO_ERR_NO_DEVICE, "device %d is %s; libvo_hip is built for gfx950 only", device,
                prop.gcnArchName);
  vo_ctx* c = new vo_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (es != hipSuccess) { delete c; return fail(VO_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(es)); }
    c->own_stream = true;
  }
  { std::lock_guard<std::mutex> lk(g_ctx_mu); c->id = g_ctx_next_id++; g_ctx_live.insert(c); }
  *out = c;
  return VO_OK;
}

int vo_ctx_destroy(vo_ctx* c) {
  if (!c) return VO_OK;
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx_live.erase(c)) return fail(VO_ERR_INVALID_ARG, "vo_ctx_destroy: not a live context (destroyed twice?)");
  }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->scratch.release(); c->best.release(); c->table.release(); c->counts.release();
  c->batch_pack.release(); c->batch_bad.release(); c->prune_ws.release(); c->epi_ws.release();
  if (c->hint_ev) (void)hipEventDestroy(c->hint_ev);
  if (c->hint_host) (void)hipHostFree(c->hint_host);
  if (c->hint_dev) (void)hipFree(c->hint_dev); c->batch_states.release(); c->batch_partials.release(); c->batch_help.release();
  for (auto& b : c->in) b.release();
  for (auto& b : c->out) b.release();
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return VO_OK;
}

int vo_ctx_synchronize(vo_ctx* c) {
  VO_REQUIRE(c, "ctx is null");
  VO_NOT_CAPTURING(c);
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  return VO_OK;
}

void* vo_ctx_stream(vo_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }
int vo_ctx_device(vo_ctx* c) { return c ? c->device : -1; }
int vo_ctx_capturing(vo_ctx* c) { return (ctx_alive(c) && c->capturing) ? 1 : 0; }
int vo_ctx_alive(vo_ctx* c) { return ctx_alive(c) ? 1 : 0; }
unsigned long long vo_ctx_id(vo_ctx* c) { std::lock_guard<std::mutex> lk(g_ctx_mu); return (c && g_ctx_live.count(c)) ? c->id : 0ull; }

int vo_ctx_device_info(vo_ctx* c, char* name, int name_len, int* n_cu) {
  VO_REQUIRE(c, "ctx is null");
  if (name && name_len > 0) { strncpy(name, c->name, (size_t)name_len - 1); name[name_len - 1] = 0; }
  if (n_cu) *n_cu = c->n_cu;
  return VO_OK;
}

int vo_ctx_begin_capture(vo_ctx* c) {
  VO_REQUIRE(c, "ctx is null");
  VO_REQUIRE(!c->capturing, "already capturing");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  c->capturing = true;
  return VO_OK;
}

int vo_ctx_end_capture(vo_ctx* c, vo_graph** out) {
  VO_REQUIRE(c && out, "null argument");
  VO_REQUIRE(c->capturing, "not capturing");
  *out = nullptr;
  c->capturing = false;
  hipGraph_t graph = nullptr;
  hipError_t ee = hipStreamEndCapture(c->stream, &graph);
  if (ee != hipSuccess) {
    // a capture invalidated by a call that cannot be captured: make sure the stream is out of capture mode again
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      if (c->own_stream) {
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) { (void)hipStreamDestroy(c->stream); c->stream = fresh; }
      }
    }
    return fail(VO_ERR_HIP, "hipStreamEndCapture: %s (the captured sequence contained a call that cannot be captured)", hipGetErrorString(ee));
  }
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(VO_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
  vo_graph* g = new vo_graph();
  g->ctx = c;
  g->ctx_id = c->id;
  g->exec = exec;
  *out = g;
  return VO_OK;
}

int vo_graph_launch(vo_graph* g) {
  VO_REQUIRE(g && g->exec, "null graph");
  VO_REQUIRE(ctx_alive(g->ctx, g->ctx_id), "the context this graph was captured on has been destroyed");
  if (int r = set_device(g->ctx)) return r;
  VO_HIP_CHECK(hipGraphLaunch(g->exec, g->ctx->stream));
  return VO_OK;
}

int vo_graph_destroy(vo_graph* g) {
  if (!g) return VO_OK;
  if (ctx_alive(g->ctx, g->ctx_id)) (void)hipStreamSynchronize(g->ctx->stream);
  else (void)hipDeviceSynchronize();
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  delete g;
  return VO_OK;
}

// ---- cross-context ordering ------------------------------------------------------------------
int vo_event_create(vo_ctx* c, vo_event** out) {
  VO_REQUIRE(c && out, "null argument");
  *out = nullptr;
  if (int r = set_device(c)) return r;
  hipEvent_t e = nullptr;
  VO_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  vo_event* v = new vo_event();
  v->device = c->device; v->ev = e;
  *out = v;
  return VO_OK;
}

int vo_event_record(vo_event* ev, vo_ctx* c) {
  VO_REQUIRE(ev && ev->ev && c, "null argument");
  VO_REQUIRE(ev->device == c->device, "event and context live on different devices");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(hipEventRecord(ev->ev, c->stream));
  return VO_OK;
}

int vo_ctx_wait_event(vo_ctx* c, vo_event* ev) {
  VO_REQUIRE(ev && ev->ev && c, "null argument");
  VO_REQUIRE(ev->device == c->device, "event and context live on different devices");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(hipStreamWaitEvent(c->stream, ev->ev, 0));
  return VO_OK;
}

int vo_event_destroy(vo_event* ev) {
  if (!ev) return VO_OK;
  if (ev->ev) (void)hipEventDestroy(ev->ev);
  delete ev;
  return VO_OK;
}

int vo_dev_alloc(vo_ctx* c, size_t bytes, void** dptr) {
  VO_REQUIRE(c && dptr, "null argument");
  VO_NOT_CAPTURING(c);
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 16));
  return VO_OK;
}

int vo_dev_free(vo_ctx* c, void* dptr) {
  VO_REQUIRE(c, "ctx is null");
  VO_NOT_CAPTURING(c);
  if (!dptr) return VO_OK;
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  VO_HIP_CHECK(hipFree(dptr));
  return VO_OK;
}

int vo_memcpy_h2d(vo_ctx* c, void* dst, const void* src, size_t bytes) {
  VO_REQUIRE(c && (bytes == 0 || (dst && src)), "null argument");
  VO_NOT_CAPTURING(c);
  if (bytes == 0) return VO_OK;
  VO_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  return VO_OK;
}

int vo_memcpy_d2h(vo_ctx* c, void* dst, const void* src, size_t bytes) {
  VO_REQUIRE(c && (bytes == 0 || (dst && src)), "null argument");
  VO_NOT_CAPTURING(c);
  if (bytes == 0) return VO_OK;
  VO_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  return VO_OK;
}

}  // extern "C"

// ---- helpers for the host-pointer entry points ------------------------------------
static int upload(vo_ctx* c, DevBuf& b, const void* src, size_t bytes) {
  VO_HIP_CHECK(b.ensure(bytes ? bytes : 16, c->stream));
  if (bytes) VO_HIP_CHECK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  return VO_OK;
}

static int ensure_counts(vo_ctx* c) {
  VO_HIP_CHECK(c->counts.ensure(64 * sizeof(int), c->stream));
  return VO_OK;
}

static int ensure_scratch(vo_ctx* c, int n) {
  VO_HIP_CHECK(c->scratch.ensure(compaction_scratch_ints(n) * sizeof(int), c->stream));
  return VO_OK;
}

static CamK make_cam(int rows, int cols, int z_near, int z_far, const float K[9]) {
  CamK cam;
  for (int i = 0; i < 9; ++i) cam.K[i] = K[i];
  cam.rows = rows; cam.cols = cols; cam.z_near = z_near; cam.z_far = z_far;
  return cam;
}

extern "C" {

// ---- projectPoints ------------------------------------------------------------------
int vo_project_points_dev(vo_ctx* c, int rows, int cols, int z_near, int z_far, const float K[9],
                          const float T[16], const float* d_world, int n, int keep_indices,
                          float* d_out_uv, int* d_counts) {
  VO_REQUIRE(c && K && T && d_counts, "null argument");
  VO_REQUIRE(n >= 0 && (n == 0 || (d_world && d_out_uv)), "bad point arrays");
  if (int r = set_device(c)) return r;
  if (int r = ensure_scratch(c, n)) return r;
  VO_HIP_CHECK(launch_project_points(c->stream, make_cam(rows, cols, z_near, z_far, K),
                                     pose_from_T16(T), d_world, n, keep_indices, d_out_uv, d_counts,
                                     c->scratch.as<int>()));
  return VO_OK;
}

int vo_project_points(vo_ctx* c, int rows, int cols, int z_near, int z_far, const float K[9],
                      const float T[16], const float* world, int n, int keep_indices, float* out_uv,
                      int* n_out, int* n_inside) {
  VO_REQUIRE(c && K && T, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n >= 0 && (n == 0 || (world && out_uv)), "bad point arrays");
  if (int r = set_device(c)) return r;
  if (int r = upload(c, c->in[0], world, sizeof(float) * 3 * (size_t)n)) return r;
  VO_HIP_CHECK(c->out[0].ensure(sizeof(float) * 2 * (size_t)(n ? n : 1), c->stream));
  if (int r = ensure_counts(c)) return r;
  if (int r = vo_project_points_dev(c, rows, cols, z_near, z_far, K, T, c->in[0].as<float>(), n,
                                    keep_indices, c->out[0].as<float>(), c->counts.as<int>()))
    return r;
  int h[2] = {0, 0};
  VO_HIP_CHECK(hipMemcpyAsync(h, c->counts.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (h[0] > 0)
    VO_HIP_CHECK(hipMemcpyAsync(out_uv, c->out[0].p, sizeof(float) * 2 * (size_t)h[0], hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (n_out) *n_out = h[0];
  if (n_inside) *n_inside = h[1];
  return VO_OK;
}

// ---- PICPSolver -----------------------------------------------------------------------
int vo_picp_create(vo_ctx* c, vo_picp** out) {
  VO_REQUIRE(c && out, "null argument");
  VO_NOT_CAPTURING(c);
  *out = nullptr;
  if (int r = set_device(c)) return r;
  vo_picp* s = new vo_picp();
  s->ctx = c;
  s->ctx_id = c->id;
  memset(&s->hp, 0, sizeof(s->hp));
  const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  s->hp.cam = make_cam(100, 100, 0, 10, I3);     // Camera defaults, camera.h:16-21
  s->hp.thr = 1000.f;                            // picp_solver.cpp:13
  s->hp.damping = 1.f;                           // picp_solver.cpp:10
  s->hp.keep_outliers = 0;
  s->hp.n_corr = 0;
  const char* g = getenv("VO_PICP_GRAPH");
  if (g) s->use_graph = atoi(g);
  if (const char* ra = getenv("VO_PICP_RUN_AHEAD")) { const int v = atoi(ra); s->run_ahead = v < 0 ? 0 : (v > PICP_SLOTS - 2 ? PICP_SLOTS - 2 : v); }
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_params), sizeof(PicpParams));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_state), sizeof(PicpState));
  if (e == hipSuccess) e = hipMemsetAsync(s->d_state, 0, sizeof(PicpState), c->stream);
  if (e != hipSuccess) {
    if (s->d_params) (void)hipFree(s->d_params);
    if (s->d_state) (void)hipFree(s->d_state);
    delete s;
    return fail(VO_ERR_HIP, "vo_picp_create: %s", hipGetErrorString(e));
  }
  // identity pose
  float pose[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  VO_HIP_CHECK(hipMemcpyAsync(s->d_state->pose[0], pose, sizeof(pose), hipMemcpyHostToDevice, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  *out = s;
  return VO_OK;
}

int vo_picp_destroy(vo_picp* s) {
  if (!s) return VO_OK;
  if (ctx_alive(s->ctx, s->ctx_id)) {
    VO_NOT_CAPTURING(s->ctx);
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
  } else {
    (void)hipDeviceSynchronize();       // the context went first: its stream is gone, the solver's memory is not
  }
  for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
  s->world_own.release(); s->meas_own.release(); s->pairs_own.release();
  s->packed.release(); s->partials.release();
  if (s->d_params) (void)hipFree(s->d_params);
  if (s->d_state) (void)hipFree(s->d_state);
  delete s;
  return VO_OK;
}

int vo_picp_set_pose(vo_picp* s, const float T[16]) {
  VO_REQUIRE(s && T, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  if (int r = set_device(s->ctx)) return r;
  const Pose P = pose_from_T16(T);
  float pose[12];
  for (int i = 0; i < 9; ++i) pose[i] = P.R[i];
  for (int i = 0; i < 3; ++i) pose[9 + i] = P.t[i];
  s->pending_T0 = nullptr;
  if (int r = picp_flush(s)) return r;     // open rounds first: H, b and the statistics stay those of the last round
  VO_HIP_CHECK(hipMemcpyAsync(s->d_state->pose[0], pose, sizeof(pose), hipMemcpyHostToDevice,
                              s->ctx->stream));
  VO_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));   // `pose` is a stack buffer
  return VO_OK;
}

int vo_picp_set_camera(vo_picp* s, int rows, int cols, int z_near, int z_far, const float K[9],
                       const float T[16]) {
  VO_REQUIRE(s && K && T, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  s->hp.cam = make_cam(rows, cols, z_near, z_far, K);
  s->params_dirty = true;
  return vo_picp_set_pose(s, T);
}

int vo_picp_set_kernel_threshold(vo_picp* s, float thr) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  s->hp.thr = thr;
  s->params_dirty = true;
  return VO_OK;
}

int vo_picp_get_kernel_threshold(vo_picp* s, float* thr) {
  VO_REQUIRE(s && thr, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  *thr = s->hp.thr;
  return VO_OK;
}

int vo_picp_set_points_dev(vo_picp* s, const float* d_world, int n_world, const float* d_meas, int n_meas) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_REQUIRE(n_world >= 0 && n_meas >= 0, "negative count");
  VO_REQUIRE((n_world == 0 || d_world) && (n_meas == 0 || d_meas), "null point array");
  s->d_world = d_world; s->n_world = n_world;
  s->d_meas = d_meas; s->n_meas = n_meas;
  s->have_points = true;
  s->packed_valid = false;
  return VO_OK;
}

int vo_picp_set_points(vo_picp* s, const float* world, int n_world, const float* meas, int n_meas) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  VO_REQUIRE(n_world >= 0 && n_meas >= 0, "negative count");
  VO_REQUIRE((n_world == 0 || world) && (n_meas == 0 || meas), "null point array");
  if (int r = set_device(s->ctx)) return r;
  if (int r = upload(s->ctx, s->world_own, world, sizeof(float) * 3 * (size_t)n_world)) return r;
  if (int r = upload(s->ctx, s->meas_own, meas, sizeof(float) * 2 * (size_t)n_meas)) return r;
  VO_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
  return vo_picp_set_points_dev(s, s->world_own.as<float>(), n_world, s->meas_own.as<float>(), n_meas);
}

}  // extern "C"

__global__ void T16_to_pose12_kernel(const float* T, float* p) {
  const int k = threadIdx.x;
  if (k < 9) p[k] = T[(k % 3) + 4 * (k / 3)];
  else if (k < 12) p[k] = T[12 + (k - 9)];
}

static int picp_prepare(vo_picp* s, const int32_t* d_pairs, int n_pairs, const int* d_n, int keep_outliers) {
  vo_ctx* c = s->ctx;
  if (!s->have_points) return fail(VO_ERR_NOT_READY, "vo_picp: set_points has not been called");
  if (s->hp.keep_outliers != (keep_outliers ? 1 : 0)) { s->hp.keep_outliers = keep_outliers ? 1 : 0; s->params_dirty = true; }
  if (s->params_dirty && c->capturing)
    return fail(VO_ERR_NOT_READY, "vo_picp: parameters changed inside a graph capture (run the sequence once before capturing)");
  if (s->params_dirty) {
    // n_corr is owned by the pack kernel: keep the device value
    VO_HIP_CHECK(hipMemcpyAsync(s->d_params, &s->hp, offsetof(PicpParams, n_corr), hipMemcpyHostToDevice,
                                c->stream));
    VO_HIP_CHECK(hipStreamSynchronize(c->stream));
    s->params_dirty = false;
  }
  if (!s->packed_valid) {
    const size_t cap = ((size_t)(n_pairs > 0 ? n_pairs : 1) + 3) & ~(size_t)3;
    VO_HIP_CHECK(s->packed.ensure(sizeof(float) * 5 * cap, c->stream));
    {
      // open rounds read the partial rows of their predecessor at the chain's grid: another grid closes the chain first
      const int g = picp_grid_for(n_pairs, c->n_cu);
      if (g != s->grid) { if (int r = picp_flush(s)) return r; }
      s->grid = g;
    }
    {
      // two buffers of round_up(grid,256) rows; rows >= grid are never written and must read as zero
      const size_t rows = ((size_t)s->grid + 255) & ~(size_t)255;
      const size_t bytes = sizeof(float) * PICP_SLOTS * rows * PICP_PSTRIDE * PICP_REPLICAS;
      if (bytes > s->partials.cap || s->grid != s->zeroed_for_grid) {
        VO_HIP_CHECK(s->partials.ensure(bytes, c->stream));
        VO_HIP_CHECK(hipMemsetAsync(s->partials.p, 0, s->partials.cap, c->stream));
        s->zeroed_for_grid = s->grid;
      }
    }
    PackedCorr pk{s->packed.as<float>(), s->packed.cap / (5 * sizeof(float)) & ~(size_t)3};
    VO_HIP_CHECK(launch_picp_pack(c->stream, d_pairs, d_n, n_pairs, s->d_world, s->n_world, s->d_meas,
                                  s->n_meas, pk, s->d_params, s->d_state, s->pending_T0));
    s->pending_T0 = nullptr;
    s->packed_valid = true;
  }
  if (s->pending_T0) {   // correspondences were cached: the pose reset needs its own (tiny) launch
    hipLaunchKernelGGL(T16_to_pose12_kernel, dim3(1), dim3(64), 0, c->stream, s->pending_T0, s->d_state->pose[0]);
    VO_HIP_CHECK(hipGetLastError());
    s->pending_T0 = nullptr;
  }
  return VO_OK;
}

static bool picp_chainable(const vo_picp* s) { return !s->exact && !s->ctx->capturing && picp_rounds_chain(s->grid); }

// rounds chain_len .. chain_len + n - 1 of the open chain; `count` = false: speculative launches the caller may have to
// repeat or will claim one by one (vo_picp_solve).  n > 1: one graph launch (captured once per starting slot and n).
static int picp_enqueue_chain_rounds(vo_picp* s, int n, bool count) {
  vo_ctx* c = s->ctx;
  PackedCorr pk{s->packed.as<float>(), s->packed.cap / (5 * sizeof(float)) & ~(size_t)3};
  float* partials = s->partials.as<float>();
  const bool pinhole = is_pinhole(s->hp.cam.K), keep = s->hp.keep_outliers != 0;
  const int first = s->chain_len;
  bool done = false;
  if (n > 1 && s->use_graph && first > 0) {
    static_assert(PICP_SLOTS < 64, "key of a window's graph: 64 * rounds + starting slot");
    auto key = std::make_tuple(-(64 * n + chain_it(first)), s->grid, (const void*)pk.base, pk.cap, (const void*)partials,
                               (pinhole ? 1 : 0) | (keep ? 2 : 0));
    auto it = s->graphs.find(key);
    if (it == s->graphs.end() && (s->graphs.size() >= 64 || (s->chain_wanted.size() < 4096 ? ++s->chain_wanted[key] : 3) < 3)) {
      n = 1;                                     // not (yet) worth a capture: this call's own round, plainly
    } else if (it == s->graphs.end()) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        hipError_t el = hipSuccess;
        for (int k = 0; k < n && el == hipSuccess; ++k)
          el = launch_picp_chain_round(c->stream, s->d_params, s->d_state, pk, partials, s->grid, chain_it(first + k), pinhole, keep);
        e = hipStreamEndCapture(c->stream, &graph);
        if (e == hipSuccess && el != hipSuccess) e = el;
      }
      if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        s->run_ahead = 0;                        // plain launches, one per call, from here on
        ++s->graph_failures;
        (void)fail(VO_OK, "vo_picp: graph capture of %d rounds ahead failed (%s); this solver enqueues one round per call", n, hipGetErrorString(e));
        n = 1;
      } else {
        it = s->graphs.emplace(key, exec).first;
      }
    }
    if (n > 1 && it != s->graphs.end()) {
      VO_HIP_CHECK(hipGraphLaunch(it->second, c->stream));
      done = true;
    }
  }
  if (!done) {
    if (!(s->use_graph && first > 0)) n = 1;     // (the first round of a chain is the kernel without the look at its predecessor)
    for (int k = 0; k < n; ++k)
      VO_HIP_CHECK(launch_picp_chain_round(c->stream, s->d_params, s->d_state, pk, partials, s->grid, chain_it(first + k), pinhole, keep));
  }
  if (count) s->chain_len += n;
  return n;                                      // rounds enqueued (>= 1); errors return through VO_HIP_CHECK as negative codes
}

// lazy: a single round may stay open (no finishing launch) -- host entry points only, whose getters close it
static int picp_enqueue(vo_picp* s, int n_iters, bool lazy = false) {
  vo_ctx* c = s->ctx;
  s->ahead = 0;                                  // whatever ran ahead of the caller is not claimed by this call
  if (lazy && n_iters == 1 && picp_chainable(s)) { const int r = picp_enqueue_chain_rounds(s, 1, true); return r < 0 ? r : VO_OK; }
  if (n_iters > 0) { if (int r = picp_flush(s)) return r; }
  PackedCorr pk{s->packed.as<float>(), s->packed.cap / (5 * sizeof(float)) & ~(size_t)3};
  float* partials = s->partials.as<float>();
  const bool pinhole = is_pinhole(s->hp.cam.K), keep = s->hp.keep_outliers != 0;
  if (s->exact) {
    VO_HIP_CHECK(launch_picp_exact(c->stream, s->d_params, s->d_state, pk, n_iters));
    return VO_OK;
  }
  if (s->use_graph && n_iters >= 2 && !c->capturing) {
    auto key = std::make_tuple(n_iters, s->grid, (const void*)pk.base, pk.cap, (const void*)partials,
                               (pinhole ? 1 : 0) | (keep ? 2 : 0));
    auto it = s->graphs.find(key);
    if (it == s->graphs.end()) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        hipError_t el = launch_picp_rounds(c->stream, s->d_params, s->d_state, pk, partials, s->grid, n_iters, pinhole, keep);
        e = hipStreamEndCapture(c->stream, &graph);
        if (e == hipSuccess && el != hipSuccess) e = el;
      }
      if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        // The rounds still run, as plain launches (same kernels, same results, ~10 % slower per round at 50k), for the rest
        // of this handle's life -- but not silently: the reason is left in vo_last_error() once and counted, so that a host
        // program (or tests/test_gpu_more.py) can see that its solver lost the graph path.
        s->use_graph = 0;
        ++s->graph_failures;
        (void)fail(VO_OK, "vo_picp: graph capture of %d rounds failed (%s); this solver now uses plain launches", n_iters,
                   hipGetErrorString(e));
      } else {
        if (s->graphs.size() >= 64) {            // (a handle used at ever new sizes: start over -- once nothing of it is in flight)
          (void)hipStreamSynchronize(c->stream);
          for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
          s->graphs.clear();
          s->chain_wanted.clear();
        }
        it = s->graphs.emplace(key, exec).first;
      }
    }
    if (s->use_graph && it != s->graphs.end()) {
      VO_HIP_CHECK(hipGraphLaunch(it->second, c->stream));
      return VO_OK;
    }
  }
  VO_HIP_CHECK(launch_picp_rounds(c->stream, s->d_params, s->d_state, pk, partials, s->grid, n_iters, pinhole, keep));
  return VO_OK;
}

extern "C" {

int vo_picp_solve_dev(vo_picp* s, const int32_t* d_pairs, int n_pairs, const int* d_n_pairs,
                      int keep_outliers, int n_iters) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_REQUIRE(n_pairs >= 0 && (n_pairs == 0 || d_pairs), "bad pairs");
  VO_REQUIRE(n_iters >= 0, "negative n_iters");
  if (int r = set_device(s->ctx)) return r;
  if (int r = picp_flush(s)) return r;     // rounds left open by vo_picp_one_round: closed before this solve's launches
  // device pairs may have been rewritten in place by the producer: always re-pack
  s->packed_valid = false;
  s->shadow_valid = false;
  if (int r = picp_prepare(s, d_pairs, n_pairs, d_n_pairs, keep_outliers)) return r;
  return picp_enqueue(s, n_iters);
}

// Uploads `pairs` unless the whole array equals what is already packed.  The reference reads the vector on every
// call (picp_solver.cpp:62): an in-place edit between two rounds must be seen, so the comparison covers every
// pair (memcmp against a host copy: ~15 us at 50k pairs, no stream synchronisation when nothing changed).
static int picp_take_pairs(vo_picp* s, const int32_t* pairs, int n_pairs) {
  vo_ctx* c = s->ctx;
  const size_t words = 2 * (size_t)n_pairs;
  const bool same = s->packed_valid && s->shadow_valid && s->shadow.size() == words &&
                    (words == 0 || memcmp(s->shadow.data(), pairs, sizeof(int32_t) * words) == 0);
  if (same) return VO_OK;
  s->shadow_valid = false;
  s->shadow.assign(pairs, pairs + words);
  if (int r = upload(c, s->pairs_own, s->shadow.data(), sizeof(int32_t) * words)) return r;
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  s->packed_valid = false;
  s->shadow_valid = true;
  return VO_OK;
}

int vo_picp_solve(vo_picp* s, const int32_t* pairs, int n_pairs, int keep_outliers, int n_iters) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  VO_REQUIRE(n_pairs >= 0 && (n_pairs == 0 || pairs), "bad pairs");
  VO_REQUIRE(n_iters >= 0, "negative n_iters");
  if (int r = set_device(s->ctx)) return r;
  // ONE round on an array that is probably the one already packed -- the reference's own loop, vo_complete.cpp:163-164:
  // a hundred oneRound calls on one vector -- is enqueued BEFORE the comparison, which then runs while the GPU works.
  // Should the array differ after all, the new pairs are uploaded and packed and the SAME round is enqueued again: the
  // speculative launch has written nothing but the partial rows and the pose slot of its own round index (from the state
  // its predecessor left, which it does not touch), and the repeat overwrites both.
  const size_t words = 2 * (size_t)n_pairs;
  if (n_iters == 1 && s->packed_valid && s->shadow_valid && s->shadow.size() == words && s->have_points &&
      s->hp.keep_outliers == (keep_outliers ? 1 : 0) && !s->params_dirty && picp_chainable(s)) {
    if (int r = picp_prepare(s, s->pairs_own.as<int32_t>(), n_pairs, nullptr, keep_outliers)) return r;   // (a pending pose only)
    int enq = 0;                                 // rounds this call put into the stream (0: its round ran ahead of it)
    if (s->ahead == 0) {
      // the caller has shown the loop (two calls in a row matched): this round and run_ahead more, one graph launch
      int want = s->streak >= 2 ? 1 + s->run_ahead : 1;
      if (s->loop_hint > s->chain_len && want > s->loop_hint - s->chain_len) want = s->loop_hint - s->chain_len;
      enq = picp_enqueue_chain_rounds(s, want, false);
      if (enq < 0) return enq;
    }
    ++s->spec_rounds;
    // (glibc's memcmp runs this pass at the L2 bandwidth of the host core -- 3.5 us for 2 x 400 KB on the EPYC 9575F of the
    // GPU box, faster than a hand-unrolled AVX2 pass: tools/micro/host_call_cost.hip)
    if (words == 0 || memcmp(s->shadow.data(), pairs, sizeof(int32_t) * words) == 0) {
      ++s->chain_len;
      if (enq > 0) s->ahead = enq - 1; else --s->ahead;
      ++s->streak;
      s->set_n = n_pairs;
      return VO_OK;
    }
    ++s->spec_redone;
  }
  s->ahead = 0;                                  // rounds that ran ahead (if any) are not this call's: round chain_len is enqueued again below
  s->streak = 0;
  if (int r = picp_take_pairs(s, pairs, n_pairs)) return r;
  // pairs_own / shadow now hold THIS array: a later vo_picp_rounds continues on it (not on a stale count from an earlier
  // vo_picp_set_correspondences, which would re-pack a mix of both arrays once set_points invalidated the packing)
  s->set_n = n_pairs;
  if (int r = picp_prepare(s, s->pairs_own.as<int32_t>(), n_pairs, nullptr, keep_outliers)) return r;
  return picp_enqueue(s, n_iters, true);
}

int vo_picp_one_round(vo_picp* s, const int32_t* pairs, int n_pairs, int keep_outliers) {
  return vo_picp_solve(s, pairs, n_pairs, keep_outliers, 1);
}

int vo_picp_set_correspondences(vo_picp* s, const int32_t* pairs, int n_pairs) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  VO_REQUIRE(n_pairs >= 0 && (n_pairs == 0 || pairs), "bad pairs");
  if (int r = set_device(s->ctx)) return r;
  s->packed_valid = false;          // explicit hand-over: always uploaded
  s->shadow_valid = false;
  if (int r = picp_take_pairs(s, pairs, n_pairs)) return r;
  s->set_n = n_pairs;
  return VO_OK;
}

int vo_picp_rounds(vo_picp* s, int keep_outliers, int n_iters) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_REQUIRE(n_iters >= 0, "negative n_iters");
  if (s->set_n < 0 || !s->shadow_valid)
    return fail(VO_ERR_NOT_READY, "vo_picp_rounds: vo_picp_set_correspondences has not been called");
  if (int r = set_device(s->ctx)) return r;
  if (int r = picp_prepare(s, s->pairs_own.as<int32_t>(), s->set_n, nullptr, keep_outliers)) return r;
  return picp_enqueue(s, n_iters, true);
}

int vo_picp_graph_info(vo_picp* s, int* use_graph, int* n_graphs, int* n_failures) {
  VO_REQUIRE(s, "null argument");
  if (use_graph) *use_graph = s->use_graph;
  if (n_graphs) *n_graphs = (int)s->graphs.size();
  if (n_failures) *n_failures = s->graph_failures;
  return VO_OK;
}

int vo_picp_chain_info(vo_picp* s, int* open_rounds, unsigned long long* speculative, unsigned long long* repeated) {
  VO_REQUIRE(s, "null argument");
  if (open_rounds) *open_rounds = s->chain_len;
  if (speculative) *speculative = s->spec_rounds;
  if (repeated) *repeated = s->spec_redone;
  return VO_OK;
}

int vo_picp_set_exact(vo_picp* s, int on) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  if ((on ? 1 : 0) != s->exact) {          // open rounds belong to the arithmetic they were enqueued in
    if (int r = set_device(s->ctx)) return r;
    if (int r = picp_flush(s)) return r;
  }
  s->exact = on ? 1 : 0;
  return VO_OK;
}

static int picp_read_state(vo_picp* s, PicpState* h) {
  if (int r = picp_flush(s)) return r;
  VO_HIP_CHECK(hipMemcpyAsync(h, s->d_state, sizeof(PicpState), hipMemcpyDeviceToHost, s->ctx->stream));
  VO_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
  if (h->n_bad > 0)
    return fail(VO_ERR_BAD_INDEX, "vo_picp: %d correspondence(s) index outside the point arrays", h->n_bad);
  return VO_OK;
}

int vo_picp_get_pose(vo_picp* s, float T[16]) {
  VO_REQUIRE(s && T, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  if (int r = set_device(s->ctx)) return r;
  PicpState h;
  const int r = picp_read_state(s, &h);
  Pose P;
  for (int i = 0; i < 9; ++i) P.R[i] = h.pose[0][i];
  for (int i = 0; i < 3; ++i) P.t[i] = h.pose[0][9 + i];
  pose_to_T16(P, T);
  return r;
}

int vo_picp_set_pose_dev(vo_picp* s, const float* d_T16) {
  VO_REQUIRE(s && d_T16, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  // consumed by the next solve: folded into its gather launch (or a 12-thread launch of its own)
  if (s->chain_len) { if (int r = set_device(s->ctx)) return r; if (int r = picp_flush(s)) return r; }
  s->pending_T0 = d_T16;
  return VO_OK;
}

int vo_picp_pose_dev_ptr(vo_picp* s, const float** d_T16) {
  VO_REQUIRE(s && d_T16, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  if (s->chain_len) { if (int r = set_device(s->ctx)) return r; if (int r = picp_flush(s)) return r; }
  *d_T16 = s->d_state->T16;
  return VO_OK;
}

__global__ void pose12_to_T16_kernel(const float* p, float* T) {
  const int k = threadIdx.x;
  if (k < 16) {
    const int r = k & 3, c = k >> 2;
    float v;
    if (r == 3) v = (c == 3) ? 1.f : 0.f;
    else if (c == 3) v = p[9 + r];
    else v = p[r + 3 * c];
    T[k] = v;
  }
}

int vo_picp_get_pose_dev(vo_picp* s, float* d_T16) {
  VO_REQUIRE(s && d_T16, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  if (int r = set_device(s->ctx)) return r;
  if (int r = picp_flush(s)) return r;
  hipLaunchKernelGGL(pose12_to_T16_kernel, dim3(1), dim3(64), 0, s->ctx->stream, s->d_state->pose[0], d_T16);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_picp_get_stats(vo_picp* s, float* chi_in, float* chi_out, int* n_in) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  if (int r = set_device(s->ctx)) return r;
  PicpState h;
  const int r = picp_read_state(s, &h);
  if (chi_in) *chi_in = h.chi_in;
  if (chi_out) *chi_out = h.chi_out;
  if (n_in) *n_in = h.n_in;
  return r;
}

#ifdef VO_STAMPS
// diagnostic build only: copies the s_memtime stamps of the last solve (128 rounds x 8 stamps)
int vo_debug_get_stamps(vo_picp* s, unsigned long long* out) {
  if (int r = picp_flush(s)) return r;
  PicpState h;
  VO_HIP_CHECK(hipMemcpyAsync(&h, s->d_state, sizeof(PicpState), hipMemcpyDeviceToHost, s->ctx->stream));
  VO_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
  memcpy(out, h.stamps, sizeof(h.stamps));
  return VO_OK;
}
#endif

int vo_picp_get_system(vo_picp* s, float H[36], float b[6]) {
  VO_REQUIRE(s, "null argument");
  VO_REQUIRE(ctx_alive(s->ctx, s->ctx_id), "the context this solver was made on has been destroyed");
  VO_NOT_CAPTURING(s->ctx);
  if (int r = set_device(s->ctx)) return r;
  PicpState h;
  const int r = picp_read_state(s, &h);
  if (H) memcpy(H, h.H, sizeof(h.H));
  if (b) memcpy(b, h.b, sizeof(h.b));
  return r;
}

int vo_picp_batch_set_form(vo_ctx* c, int form) {
  VO_REQUIRE(c, "ctx is null");
  VO_REQUIRE(form >= 0 && form <= 3,
             "form must be 0 (auto), 1 (one launch per round), 2 (one workgroup per problem) or 3 (reference-order arithmetic)");
  c->batch_form = form;
  return VO_OK;
}

int vo_picp_batch_info(vo_ctx* c, int* form, int* workgroups) {
  VO_REQUIRE(c, "ctx is null");
  if (form) *form = c->batch_last_form;
  if (workgroups) *workgroups = c->batch_last_wgs;
  return VO_OK;
}

static int picp_solve_batch(vo_ctx* c, int n_problems, int rows, int cols, int z_near, int z_far,
                            const float K[9], float thr, int keep_outliers, const float* d_world,
                            size_t world_stride, const float* d_meas, size_t meas_stride,
                            const int32_t* d_pairs, size_t pairs_stride, const int* d_n_pairs,
                            const float* d_T0, int n_iters, float* d_T_out, float* d_stats_out, const float* d_X_world);
// the same in two steps: everything up to (not including) the launches -- workspaces sized, BatchArgs filled, the dropped-
// correspondence counters zeroed on the stream -- so that a caller can hand the arguments to the join as its gather sink
static int picp_batch_prepare(vo_ctx* c, int n_problems, int rows, int cols, int z_near, int z_far,
                              const float K[9], float thr, int keep_outliers, const float* d_world,
                              size_t world_stride, const float* d_meas, size_t meas_stride,
                              const int32_t* d_pairs, size_t pairs_stride, const int* d_n_pairs,
                              const float* d_T0, int n_iters, float* d_T_out, float* d_stats_out, const float* d_X_world,
                              BatchArgs& a);

int vo_picp_solve_batch_dev(vo_ctx* c, int n_problems, int rows, int cols, int z_near, int z_far,
                            const float K[9], float thr, int keep_outliers, const float* d_world,
                            size_t world_stride, const float* d_meas, size_t meas_stride,
                            const int32_t* d_pairs, size_t pairs_stride, const int* d_n_pairs,
                            const float* d_T0, int n_iters, float* d_T_out, float* d_stats_out) {
  return picp_solve_batch(c, n_problems, rows, cols, z_near, z_far, K, thr, keep_outliers, d_world, world_stride, d_meas, meas_stride,
                          d_pairs, pairs_stride, d_n_pairs, d_T0, n_iters, d_T_out, d_stats_out, nullptr);
}

// d_X_world (n_problems x 16, or null): rigid transform applied to every world point the gather fetches
static int picp_solve_batch(vo_ctx* c, int n_problems, int rows, int cols, int z_near, int z_far,
                            const float K[9], float thr, int keep_outliers, const float* d_world,
                            size_t world_stride, const float* d_meas, size_t meas_stride,
                            const int32_t* d_pairs, size_t pairs_stride, const int* d_n_pairs,
                            const float* d_T0, int n_iters, float* d_T_out, float* d_stats_out, const float* d_X_world) {
  BatchArgs a;
  if (int r = picp_batch_prepare(c, n_problems, rows, cols, z_near, z_far, K, thr, keep_outliers, d_world, world_stride, d_meas,
                                 meas_stride, d_pairs, pairs_stride, d_n_pairs, d_T0, n_iters, d_T_out, d_stats_out, d_X_world, a))
    return r;
  if (n_problems == 0) return VO_OK;
  VO_HIP_CHECK(launch_picp_batch(c->stream, a));
  return VO_OK;
}

static int picp_batch_prepare(vo_ctx* c, int n_problems, int rows, int cols, int z_near, int z_far,
                              const float K[9], float thr, int keep_outliers, const float* d_world,
                              size_t world_stride, const float* d_meas, size_t meas_stride,
                              const int32_t* d_pairs, size_t pairs_stride, const int* d_n_pairs,
                              const float* d_T0, int n_iters, float* d_T_out, float* d_stats_out, const float* d_X_world,
                              BatchArgs& a) {
  VO_REQUIRE(c && K, "null argument");
  VO_REQUIRE(n_problems >= 0 && n_iters >= 0, "negative count");
  VO_REQUIRE(n_problems <= 65535, "more than 65535 problems per call (the problem is a grid dimension)");
  if (n_problems == 0) return VO_OK;
  VO_REQUIRE(d_world && d_meas && d_pairs && d_n_pairs && d_T_out, "null device array");
  VO_REQUIRE(world_stride > 0 && meas_stride > 0 && pairs_stride > 0, "zero stride");
  VO_REQUIRE(world_stride < 0x7fffffff && meas_stride < 0x7fffffff && pairs_stride < 0x7fffffff, "stride too large");
  if (int r = set_device(c)) return r;
  a.prepacked = 0;
  a.cam = make_cam(rows, cols, z_near, z_far, K);
  a.thr = thr; a.damping = 1.f; a.keep_outliers = keep_outliers ? 1 : 0;
  a.n_iters = n_iters; a.n_problems = n_problems;
  a.world = d_world; a.world_stride = world_stride;
  a.meas = d_meas; a.meas_stride = meas_stride;
  a.pairs = d_pairs; a.pairs_stride = pairs_stride;
  a.n_pairs = d_n_pairs; a.T0 = d_T0; a.T_out = d_T_out; a.stats_out = d_stats_out;
  a.cap = (pairs_stride + 3) & ~(size_t)3;
  a.n_world = (int)world_stride; a.n_meas = (int)meas_stride;
  VO_HIP_CHECK(c->batch_pack.ensure(sizeof(float) * 5 * a.cap * (size_t)n_problems, c->stream));
  a.packed = c->batch_pack.as<float>();
  a.n_bad = nullptr;
  if (d_stats_out) {
    VO_HIP_CHECK(c->batch_bad.ensure(sizeof(int) * (size_t)n_problems, c->stream));
    VO_HIP_CHECK(hipMemsetAsync(c->batch_bad.p, 0, sizeof(int) * (size_t)n_problems, c->stream));
    a.n_bad = c->batch_bad.as<int>();
  }
  a.states = nullptr; a.partials = nullptr; a.params = nullptr; a.grid = 0; a.exact = 0;
  a.X_world = d_X_world;
  static const int env_form = [] { const char* e = getenv("VO_PICP_BATCH_FORM"); return e ? atoi(e) : 0; }();
  const int form = c->batch_form ? c->batch_form : env_form;
  a.exact = form == 3;
  const bool rounds = form == 1 || (form == 0 && picp_batch_prefers_rounds(n_problems, a.cap, n_iters, c->n_cu));
  if (rounds && !c->capturing) {
    // a few problems: one launch per round with many workgroups per problem (the single-problem kernel with the
    // problem as a grid dimension) instead of one workgroup per problem
    a.grid = picp_grid_for((int)a.cap, c->n_cu);
    const size_t rows = ((size_t)a.grid + 255) & ~(size_t)255;
    const size_t part_bytes = sizeof(float) * PICP_SLOTS * rows * PICP_PSTRIDE * PICP_REPLICAS * (size_t)n_problems;
    const size_t states_cap = c->batch_states.cap;
    VO_HIP_CHECK(c->batch_states.ensure(sizeof(PicpState) * (size_t)n_problems + sizeof(PicpParams), c->stream));
    if (c->batch_states.cap != states_cap) c->batch_params_dev = nullptr;               // reallocated: nothing uploaded yet
    VO_HIP_CHECK(c->batch_partials.ensure(part_bytes, c->stream));
    VO_HIP_CHECK(hipMemsetAsync(c->batch_partials.p, 0, part_bytes, c->stream));       // rows >= grid must read as zero
    PicpParams hp;
    memset(&hp, 0, sizeof(hp));
    hp.cam = a.cam; hp.thr = thr; hp.damping = 1.f; hp.keep_outliers = a.keep_outliers; hp.n_corr = 0;
    PicpParams* d_params = reinterpret_cast<PicpParams*>(c->batch_states.as<char>() + sizeof(PicpState) * (size_t)n_problems);
    if (d_params != c->batch_params_dev || memcmp(&hp, &c->batch_params_host, sizeof(hp)) != 0) {
      // the parameter block changes rarely: upload (and wait, the source is this object) only then
      c->batch_params_host = hp;
      VO_HIP_CHECK(hipMemcpyAsync(d_params, &c->batch_params_host, sizeof(hp), hipMemcpyHostToDevice, c->stream));
      VO_HIP_CHECK(hipStreamSynchronize(c->stream));
      c->batch_params_dev = d_params;
    }
    a.states = c->batch_states.as<PicpState>();
    a.partials = c->batch_partials.as<float>();
    a.params = d_params;
  }
  picp_help_args(a, nullptr, 0);
  if (!a.states && !a.exact && picp_batch_shares(n_problems, a.cap, n_iters, c->n_cu)) {
    // one workgroup per problem would leave CUs without one: their waves take work off the others (picp_batch_shared_kernel)
    VO_HIP_CHECK(c->batch_help.ensure(sizeof(unsigned long long) * picp_help_words(n_problems, c->n_cu), c->stream));
    picp_help_args(a, c->batch_help.as<unsigned long long>(), c->n_cu);
  }
  c->batch_last_form = a.exact ? 3 : a.states ? 1 : a.help_words ? 4 : 2;
  c->batch_last_wgs = a.states ? a.grid * n_problems : a.help_words ? a.help_grid : n_problems;
  return VO_OK;
}

// ---- matcher ------------------------------------------------------------------------------
// the sorted variants pay several small launches: worth it from ~4 M candidate pairs on
static int match_workspace(vo_ctx* c, int variant, int nt, int nq, int n_frames, void** ws) {
  *ws = nullptr;
  if (variant == 1) return VO_OK;
  VO_HIP_CHECK(c->prune_ws.ensure(match_workspace_bytes(variant, nt, nq, n_frames), c->stream));
  *ws = c->prune_ws.p;
  return VO_OK;
}

// Auto: full scan for small sets; for one frame the bucket-pruned scan (its LDS-tiled scan is the shorter
// dependency chain when the GPU is not full: 82 vs 130 us at 50k x 50k); for many frames per call the cell-hash
// search (25 instead of ~1400 candidates per query: 4.0 vs 4.4 ms per 200 frames, its random accesses hidden
// by occupancy).  In front of either sorted search auto mode runs the exact-duplicate pass (variants 4 / 5, match.hip
// "hash-first": appearances are copied from frame to frame, so almost every query has a bitwise copy in the tree, which
// is its nearest neighbour at distance 0; the search then only sees the queries without one) -- from 8 frames per call on.
// For one frame the pass (three launches, ~45 us at 50k) pays only when NO query is left over: 55 against 78 us; a tracking
// sequence brings new landmarks with every frame, and the sorted search it then still needs costs what it costs without
// the pass (the tree must be sorted for a single open query): 123 against 75 us per frame of the synthetic 200 x 50k
// sequence.  Modes 4 / 5 ask for the pass at any size.
// VO_MATCH_AUTO=2|3 forces one of the sorted variants in auto mode, VO_MATCH_HASH=0 leaves the exact-duplicate pass out,
// VO_MATCH_HASH=1 puts it in front of every sorted search.
static int with_hash(int v, int nt, int n_frames) {
  static const int env = [] { const char* e = getenv("VO_MATCH_HASH"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  const bool on = env < 0 ? n_frames >= 8 : env == 1;
  return (on && (v == 2 || v == 3) && match_hash_supported(nt, n_frames)) ? v + 2 : v;
}
static int match_auto_flag(const vo_ctx* c) { return c->match_mode == 0 ? MATCH_VARIANT_AUTO : 0; }

// the two halves of the steering described at vo_ctx::hint_host, around a batched matcher call in automatic mode
constexpr int HINT_SKIP = 16;
static int match_hint_before(vo_ctx* c, int variant, bool* skipped) {
  static const bool off = [] { const char* e = getenv("VO_MATCH_HINT"); return e && e[0] == '0'; }();
  *skipped = false;
  if (off || c->match_mode != 0 || variant < 4 || c->capturing) return variant;
  if (c->hint_pending && hipEventQuery(c->hint_ev) == hipSuccess) {
    c->hint_pending = false;
    if (c->hint_of_skipped_call) {
      if (*c->hint_host != 0) { c->hint_skip_left = 0; c->hint_none_streak = 0; }          // copies are back: the pass again
    } else if (*c->hint_host == 0) {
      if (++c->hint_none_streak >= 2) c->hint_skip_left = HINT_SKIP;                       // twice in a row no frame took the pass: leave it out
    } else {
      c->hint_none_streak = 0;
    }
  }
  (void)hipGetLastError();                        // (hipErrorNotReady of the query is not an error of this call)
  if (c->hint_skip_left > 0) { --c->hint_skip_left; *skipped = true; return variant - 2; }      // the plain search: same pairs
  return variant;
}
// skipped: the call ran without the pass (match_hint_before took it out); best / nq_cap / sizes: its keys
static void match_hint_after(vo_ctx* c, int variant, bool skipped, const void* ws, int n_frames, const unsigned long long* d_best,
                             size_t best_stride, int nq_cap, const int* d_n1, const int* d_n2) {
  if (c->match_mode != 0 || (variant < 4 && !skipped) || c->capturing || !ws || nq_cap <= 0) return;
  if (!c->hint_host) {
    if (hipHostMalloc(reinterpret_cast<void**>(&c->hint_host), 64, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->hint_dev), 64) != hipSuccess ||
        hipEventCreateWithFlags(&c->hint_ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (c->hint_host) { (void)hipHostFree(c->hint_host); c->hint_host = nullptr; }      // no steering, nothing else changes
      return;
    }
    *c->hint_host = 1;
  }
  if (c->hint_pending) return;                     // an answer is still on its way
  const hipError_t el = skipped ? launch_match_hint_from_best(c->stream, d_best, best_stride, nq_cap, d_n1, d_n2, n_frames, c->hint_dev)
                                : launch_match_hint(c->stream, ws, n_frames, c->hint_dev);
  c->hint_of_skipped_call = skipped;
  if (el != hipSuccess ||
      hipMemcpyAsync(c->hint_host, c->hint_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipEventRecord(c->hint_ev, c->stream) != hipSuccess) { (void)hipGetLastError(); return; }
  c->hint_pending = true;
}
static int match_variant(const vo_ctx* c, int nt, int nq, int n_frames) {
  const bool cells_ok = match_cells_supported(nt, nq);      // the cell-hash search serves sets of up to 1.8 M points
  if (c->match_mode != 0) {
    int m = c->match_mode;
    const bool hash = m >= 4;                               // 4 / 5: exact-duplicate pass first, at any size it takes
    if (hash) m -= 2;
    if (m == 3 && !cells_ok) m = 2;
    return (hash && match_hash_supported(nt, n_frames)) ? m + 2 : m;
  }
  // the full scan: below ~4 M candidate pairs per frame whatever the frame count, and -- a call of one or a few frames, whose
  // sorted searches are a chain of ~8 small launches (45 us) -- up to 1e8 pairs in the whole call: one frame of up to ~10 000 x
  // 10 000 points (measured, one frame, full scan against the sorted searches: 5000: 30 / 46 us, 8500: 36 / 48, 10 000: 42 / 49,
  // 12 000: 60 / 49; tools/match_sizes.py)
  {
    const double pairs = (double)nt * (double)nq;
    // (the 1e8 rule is measured for ONE frame only; calls of 2..7 frames keep the 4 M-pairs-per-frame crossover)
    if (pairs < 4.0e6 || (n_frames <= 1 && pairs < 1.0e8)) return 1;
  }
  static const int forced = [] { const char* e = getenv("VO_MATCH_AUTO"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 3) ? v : 0; }();
  if (forced) return with_hash((forced == 3 && !cells_ok) ? 2 : forced, nt, n_frames);
  return with_hash((n_frames >= 8 && cells_ok) ? 3 : 2, nt, n_frames);
}

// frames of different sizes: the full scan, or -- from the sizes on where a sorted search pays -- the cell-hash search (its
// workspace is then laid out for the larger capacity in both roles), with the exact-duplicate pass in front of it like
// match_variant decides; the bucket-pruned scan takes one size only
static int ragged_variant(const vo_ctx* c, int nt_cap, int q_cap, int n_frames) {
  const int v = match_variant(c, nt_cap, q_cap, n_frames);
  if (v == 1 || !match_cells_supported(nt_cap, nt_cap)) return 1;
  return v >= 4 ? 5 : 3;
}

int vo_match_set_mode(vo_ctx* c, int mode) {
  VO_REQUIRE(c, "ctx is null");
  VO_REQUIRE(mode >= 0 && mode <= 5, "mode must be 0 (auto), 1 (full scan), 2 (bucket-pruned scan), 3 (cell-hash search), "
                                     "4 / 5 (exact-duplicate pass, then 2 / 3)");
  c->match_mode = mode;
  c->hint_skip_left = 0;              // what the automatic mode learnt from earlier calls is forgotten
  c->hint_none_streak = 0;
  c->hint_pending = false;
  return VO_OK;
}

int vo_match_appearances_dev(vo_ctx* c, const float* d_a1, int n1, const float* d_a2, int n2,
                             float radius, int32_t* d_out_pairs, int* d_n_out) {
  VO_REQUIRE(c && d_n_out, "null argument");
  VO_REQUIRE(n1 >= 0 && n2 >= 0, "negative count");
  VO_REQUIRE((n1 == 0 || d_a1) && (n2 == 0 || d_a2), "null appearance array");
  VO_REQUIRE(aligned8(d_a1, d_a2, d_out_pairs), "device array not on an 8-byte boundary");
  const int nq = n1 < n2 ? n1 : n2;
  VO_REQUIRE(nq == 0 || d_out_pairs, "null output");
  if (int r = set_device(c)) return r;
  if (int r = ensure_scratch(c, nq)) return r;
  VO_HIP_CHECK(c->best.ensure(sizeof(unsigned long long) * (size_t)(nq ? nq : 1), c->stream));
  // the pruned scan pays ~8 small launches of sorting: worth it from ~4 M candidate pairs on
  const int nt = n1 > n2 ? n1 : n2;
  const int variant = match_variant(c, nt, nq, 1);
  void* ws = nullptr;
  if (nq > 0) if (int r = match_workspace(c, variant, nt, nq, 1, &ws)) return r;
  VO_HIP_CHECK(launch_match(c->stream, d_a1, n1, d_a2, n2, radius, d_out_pairs, d_n_out,
                            c->best.as<unsigned long long>(), c->scratch.as<int>(), c->n_cu, ws, variant | match_auto_flag(c)));
  return VO_OK;
}

int vo_match_appearances(vo_ctx* c, const float* a1, int n1, const float* a2, int n2, float radius,
                         int32_t* out_pairs, int* n_out) {
  VO_REQUIRE(c && n_out, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n1 >= 0 && n2 >= 0, "negative count");
  VO_REQUIRE((n1 == 0 || a1) && (n2 == 0 || a2), "null appearance array");
  VO_REQUIRE((n1 == 0 || n2 == 0) || out_pairs, "null output");
  if (int r = set_device(c)) return r;
  const int nq = n1 < n2 ? n1 : n2;
  if (int r = upload(c, c->in[0], a1, sizeof(float) * 10 * (size_t)n1)) return r;
  if (int r = upload(c, c->in[1], a2, sizeof(float) * 10 * (size_t)n2)) return r;
  VO_HIP_CHECK(c->out[0].ensure(sizeof(int32_t) * 2 * (size_t)(nq ? nq : 1), c->stream));
  if (int r = ensure_counts(c)) return r;
  if (int r = vo_match_appearances_dev(c, c->in[0].as<float>(), n1, c->in[1].as<float>(), n2, radius,
                                       c->out[0].as<int32_t>(), c->counts.as<int>()))
    return r;
  int h = 0;
  VO_HIP_CHECK(hipMemcpyAsync(&h, c->counts.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (h > 0) VO_HIP_CHECK(hipMemcpyAsync(out_pairs, c->out[0].p, sizeof(int32_t) * 2 * (size_t)h, hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  *n_out = h;
  return VO_OK;
}

// ---- many frame pairs' appearances at once, every pair its own sizes -------------------------
int vo_match_appearances_batch_dev(vo_ctx* c, int n_frames, const float* d_a1, int cap1, const int* d_n1, const float* d_a2,
                                   int cap2, const int* d_n2, float radius, int32_t* d_out_pairs, int* d_n_out) {
  VO_REQUIRE(c && d_n_out, "null argument");
  VO_REQUIRE(n_frames >= 0 && cap1 >= 0 && cap2 >= 0, "negative count");
  if (n_frames == 0) return VO_OK;
  VO_REQUIRE(n_frames <= 65535, "more than 65535 frames per call (the frame is a grid dimension)");
  VO_REQUIRE((d_n1 == nullptr) == (d_n2 == nullptr), "per-frame sizes: give both arrays or neither");
  const int q = cap1 < cap2 ? cap1 : cap2;
  VO_REQUIRE((cap1 == 0 || d_a1) && (cap2 == 0 || d_a2) && (q == 0 || d_out_pairs), "null device array");
  VO_REQUIRE(aligned8(d_a1, d_a2, d_out_pairs), "device array not on an 8-byte boundary");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(c->scratch.ensure(sizeof(int) * compaction_scratch_ints(q) * (size_t)n_frames, c->stream));
  VO_HIP_CHECK(c->best.ensure(sizeof(unsigned long long) * (size_t)(q ? q : 1) * (size_t)n_frames, c->stream));
  const int nt = cap1 > cap2 ? cap1 : cap2;
  bool skipped = false;
  const int variant_rule = d_n1 ? ragged_variant(c, nt, q, n_frames) : match_variant(c, nt, q, n_frames);
  const int variant = match_hint_before(c, variant_rule, &skipped);
  // (sized for what the rule picks, also while the steering leaves the pass out: the call that brings it back must not have to
  // grow the workspace -- a device allocation of gigabytes and a synchronisation in the middle of a run)
  void* ws = nullptr;
  if (q > 0) if (int r = match_workspace(c, variant_rule, nt, d_n1 ? nt : q, n_frames, &ws)) return r;
  VO_HIP_CHECK(launch_match_batch(c->stream, d_a1, cap1, 10 * (size_t)cap1, d_a2, cap2, 10 * (size_t)cap2, radius, d_out_pairs,
                                  (size_t)q, d_n_out, c->best.as<unsigned long long>(), c->scratch.as<int>(), c->n_cu, ws, n_frames,
                                  variant | match_auto_flag(c), d_n1, d_n2));
  match_hint_after(c, variant, skipped, ws, n_frames, c->best.as<unsigned long long>(), (size_t)q, q, d_n1, d_n2);
  return VO_OK;
}

// ---- batched frames -----------------------------------------------------------------------
static int frames_batch(vo_ctx* c, const vo_frame_batch* b, const vo_frame_sizes* sz);

int vo_frames_batch_dev(vo_ctx* c, const vo_frame_batch* b) { return frames_batch(c, b, nullptr); }

int vo_frames_batch_ragged_dev(vo_ctx* c, const vo_frame_batch* b, const vo_frame_sizes* sizes) {
  VO_REQUIRE(sizes, "null argument");
  VO_REQUIRE(sizes->n_ref && sizes->n_cur && sizes->n_model_pairs, "null per-frame size array");
  return frames_batch(c, b, sizes);
}

}  // extern "C"

static int frames_batch(vo_ctx* c, const vo_frame_batch* b, const vo_frame_sizes* sz) {
  VO_REQUIRE(c && b, "null argument");
  const int F = b->n_frames;
  VO_REQUIRE(F >= 0 && b->n_ref >= 0 && b->n_cur >= 0 && b->n_model >= 0 && b->n_model_pairs >= 0 && b->n_iters >= 0,
             "negative count");
  if (F == 0) return VO_OK;
  VO_REQUIRE(F <= 65535, "more than 65535 frames per call (the frame is a grid dimension)");
  const int q = b->n_ref < b->n_cur ? b->n_ref : b->n_cur;
  VO_REQUIRE(q > 0 && b->n_model > 0, "empty frames");
  VO_REQUIRE(b->ref_app && b->cur_app && b->ref_pts && b->cur_pts && b->model && b->model_pairs, "null input array");
  VO_REQUIRE(b->matches && b->joined && b->poses && b->tri_xyz && b->tri_pairs && b->counts, "null output array");
  VO_REQUIRE(aligned8(b->ref_app, b->cur_app, b->ref_pts, b->cur_pts, b->model_pairs, b->matches, b->joined, b->tri_pairs, b->tri_app),
             "device array not on an 8-byte boundary");
  if (int r = set_device(c)) return r;
  const int nt = b->n_ref > b->n_cur ? b->n_ref : b->n_cur;
  VO_HIP_CHECK(c->scratch.ensure(triangulate_scratch_bytes(q, F), c->stream));    // (covers the counts of the other compactions)
  VO_HIP_CHECK(c->best.ensure(sizeof(unsigned long long) * (size_t)q * (size_t)F, c->stream));
  VO_HIP_CHECK(c->table.ensure(sizeof(unsigned long long) * (size_t)(b->n_ref ? b->n_ref : 1) * (size_t)F, c->stream));
  // ragged frames (sz): the counts of the struct are capacities (= strides), frame f holds sz->n_ref[f] / n_cur[f] points and
  // n_model_pairs[f] model pairs; the matcher (full scan or cell-hash search) picks every frame's roles itself (vo_complete.cpp:15-20)
  bool skipped = false;
  const int variant_rule = sz ? ragged_variant(c, nt, q, F) : match_variant(c, nt, q, F);
  const int variant = match_hint_before(c, variant_rule, &skipped);
  void* ws = nullptr;                                     // (sized for the rule's pick: see vo_match_appearances_batch_dev)
  if (int r = match_workspace(c, variant_rule, nt, sz ? nt : q, F, &ws)) return r;
  int* n_match = b->counts;
  int* n_join = b->counts + F;
  int* n_tri = b->counts + 2 * (size_t)F;
  // compute_correspondences_images, all frames                                  vo_complete.cpp:156
  VO_HIP_CHECK(launch_match_batch(c->stream, b->ref_app, b->n_ref, 10 * (size_t)b->n_ref, b->cur_app, b->n_cur,
                                  10 * (size_t)b->n_cur, b->radius, b->matches, (size_t)q, n_match,
                                  c->best.as<unsigned long long>(), c->scratch.as<int>(), c->n_cu, ws, F, variant | match_auto_flag(c),
                                  sz ? sz->n_ref : nullptr, sz ? sz->n_cur : nullptr));
  match_hint_after(c, variant, skipped, ws, F, c->best.as<unsigned long long>(), (size_t)q, q, sz ? sz->n_ref : nullptr, sz ? sz->n_cur : nullptr);
  // X_curr * triangulated_pc                                                    vo_complete.cpp:159
  // The moved cloud as an output is optional: without it the solver's gather applies X_prev to the points it fetches (the
  // same arithmetic, PointCloud.h:80) and the pass that writes n_model points per frame only to re-read the joined ones
  // is gone (117 MB written + 37 us per 200 x 50k frames).
  const float* world = b->model;
  const float* X_world = nullptr;
  if (b->model_moved) {
    VO_HIP_CHECK(launch_transform_batch(c->stream, b->X_prev, b->model, b->n_model, (size_t)b->n_model, b->model_moved, F));
    world = b->model_moved;
  } else {
    X_world = b->X_prev;                              // (null: identity -- the points are used as they are)
  }
  // solver.init(identity) + n rounds: arguments first ...                       vo_complete.cpp:161-164
  BatchArgs pa;
  if (int r = picp_batch_prepare(c, F, b->rows, b->cols, b->z_near, b->z_far, b->K, b->kernel_threshold,
                                 b->keep_outliers, world, (size_t)b->n_model, b->cur_pts, (size_t)b->n_cur,
                                 b->joined, (size_t)q, n_join, nullptr, b->n_iters, b->poses, b->stats, X_world, pa))
    return r;
  // extract_correspondences_world                                               vo_complete.cpp:157
  // ... because the join's writing pass gathers every pair it emits straight into the solver's packed arrays (the one-
  // workgroup-per-problem and reference-order forms; the launch-per-round form of a few frames keeps its own gather pass,
  // which also sets the problems' states up): one launch and one read of the joined pairs less (0.145 + 0.038 -> ~0.15 ms
  // per 200 x 50k frames)
  static const bool fuse_on = [] { const char* e = getenv("VO_JOIN_GATHER"); return !(e && e[0] == '0'); }();
  const bool fuse = fuse_on && pa.states == nullptr && join_fuses_gather(q, b->n_model_pairs, b->n_ref);
  if (fuse) pa.prepacked = 1;
  VO_HIP_CHECK(launch_join_batch(c->stream, b->matches, q, n_match, b->model_pairs, b->n_model_pairs,
                                 sz ? sz->n_model_pairs : nullptr, b->n_ref,
                                 b->joined, n_join, c->table.as<unsigned long long>(), c->scratch.as<int>(), F, (size_t)q,
                                 (size_t)b->n_model_pairs, (size_t)q, fuse ? &pa : nullptr));
  VO_HIP_CHECK(launch_picp_batch(c->stream, pa));
  // triangulate_points with the new pose                                        vo_complete.cpp:172-173
  VO_HIP_CHECK(launch_triangulate_batch(c->stream, b->K, nullptr, b->poses, b->matches, q, n_match, b->ref_pts, b->n_ref,
                                        b->cur_pts, b->n_cur, b->tri_app ? b->cur_app : nullptr, b->tri_xyz, b->tri_pairs,
                                        b->tri_app, n_tri, c->scratch.as<int>(), F, (size_t)q, (size_t)b->n_ref,
                                        (size_t)b->n_cur, (size_t)q));
  return VO_OK;
}

extern "C" {

// ---- join ---------------------------------------------------------------------------------
int vo_join_correspondences_dev(vo_ctx* c, const int32_t* d_img, int n_img, const int* d_n_img,
                                const int32_t* d_world, int n_world, const int* d_n_world, int n_ref,
                                int32_t* d_out, int* d_n_out) {
  VO_REQUIRE(c && d_n_out, "null argument");
  VO_REQUIRE(n_img >= 0 && n_world >= 0 && n_ref >= 0, "negative count");
  VO_REQUIRE((n_img == 0 || (d_img && d_out)) && (n_world == 0 || d_world), "null pair array");
  VO_REQUIRE(aligned8(d_img, d_world, d_out), "device array not on an 8-byte boundary");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(c->scratch.ensure(join_scratch_bytes(n_img, 1), c->stream));
  VO_HIP_CHECK(c->table.ensure(sizeof(unsigned long long) * (size_t)(n_ref ? n_ref : 1), c->stream));
  VO_HIP_CHECK(launch_join(c->stream, d_img, n_img, d_n_img, d_world, n_world, d_n_world, n_ref, d_out,
                           d_n_out, c->table.as<unsigned long long>(), c->scratch.as<int>()));
  return VO_OK;
}

int vo_join_correspondences(vo_ctx* c, const int32_t* img, int n_img, const int32_t* world, int n_world,
                            int32_t* out, int* n_out) {
  VO_REQUIRE(c && n_out, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n_img >= 0 && n_world >= 0, "negative count");
  VO_REQUIRE((n_img == 0 || (img && out)) && (n_world == 0 || world), "null pair array");
  if (int r = set_device(c)) return r;
  // the reference compares arbitrary ints; negative reference indices never match
  // anything meaningful, they are treated as "no partner".
  int n_ref = 0;
  for (int j = 0; j < n_world; ++j) if (world[2 * j] >= n_ref) n_ref = world[2 * j] + 1;
  if (int r = upload(c, c->in[0], img, sizeof(int32_t) * 2 * (size_t)n_img)) return r;
  if (int r = upload(c, c->in[1], world, sizeof(int32_t) * 2 * (size_t)n_world)) return r;
  VO_HIP_CHECK(c->out[0].ensure(sizeof(int32_t) * 2 * (size_t)(n_img ? n_img : 1), c->stream));
  if (int r = ensure_counts(c)) return r;
  if (int r = vo_join_correspondences_dev(c, c->in[0].as<int32_t>(), n_img, nullptr, c->in[1].as<int32_t>(),
                                          n_world, nullptr, n_ref, c->out[0].as<int32_t>(), c->counts.as<int>()))
    return r;
  int h = 0;
  VO_HIP_CHECK(hipMemcpyAsync(&h, c->counts.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (h > 0) VO_HIP_CHECK(hipMemcpyAsync(out, c->out[0].p, sizeof(int32_t) * 2 * (size_t)h, hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  *n_out = h;
  return VO_OK;
}

// ---- radius search (fullSearch) ----------------------------------------------------------------
int vo_radius_search_dev(vo_ctx* c, const float* d_tree, int n_tree, const float* d_qry, int n_q, float radius,
                         int32_t* d_offsets, int32_t* d_indices, int capacity) {
  VO_REQUIRE(c && d_offsets, "null argument");
  VO_REQUIRE(n_tree >= 0 && n_q >= 0 && capacity >= 0, "negative count");
  VO_REQUIRE((n_tree == 0 || d_tree) && (n_q == 0 || d_qry) && (capacity == 0 || d_indices), "null array");
  VO_REQUIRE(match_cells_supported(n_tree, n_q), "more than 1 835 008 points in one set");
  VO_REQUIRE(aligned8(d_tree, d_qry), "device array not on an 8-byte boundary");       // (the cell-hash path loads rows as 8-byte words)
  if (int r = set_device(c)) return r;
  void* ws = nullptr;
  if (n_tree > 0 && n_q > 0) if (int r = match_workspace(c, 3, n_tree, n_q, 1, &ws)) return r;
  VO_HIP_CHECK(launch_radius_search(c->stream, d_tree, n_tree, d_qry, n_q, radius, d_offsets, d_indices, capacity, ws));
  return VO_OK;
}

int vo_radius_search(vo_ctx* c, const float* tree, int n_tree, const float* qry, int n_q, float radius,
                     int32_t* offsets, int32_t* indices, int capacity, int* n_total) {
  VO_REQUIRE(c && offsets && n_total, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n_tree >= 0 && n_q >= 0 && capacity >= 0, "negative count");
  VO_REQUIRE((n_tree == 0 || tree) && (n_q == 0 || qry) && (capacity == 0 || indices), "null array");
  if (int r = set_device(c)) return r;
  if (int r = upload(c, c->in[0], tree, sizeof(float) * 10 * (size_t)n_tree)) return r;
  if (int r = upload(c, c->in[1], qry, sizeof(float) * 10 * (size_t)n_q)) return r;
  VO_HIP_CHECK(c->out[0].ensure(sizeof(int32_t) * ((size_t)n_q + 1), c->stream));
  VO_HIP_CHECK(c->out[1].ensure(sizeof(int32_t) * (size_t)(capacity ? capacity : 1), c->stream));
  if (int r = vo_radius_search_dev(c, c->in[0].as<float>(), n_tree, c->in[1].as<float>(), n_q, radius,
                                   c->out[0].as<int32_t>(), c->out[1].as<int32_t>(), capacity))
    return r;
  VO_HIP_CHECK(hipMemcpyAsync(offsets, c->out[0].p, sizeof(int32_t) * ((size_t)n_q + 1), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  *n_total = offsets[n_q];
  if (*n_total > capacity)
    return fail(VO_ERR_INVALID_ARG, "vo_radius_search: %d hits, room for %d", *n_total, capacity);
  if (*n_total > 0)
    VO_HIP_CHECK(hipMemcpyAsync(indices, c->out[1].p, sizeof(int32_t) * (size_t)*n_total, hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  return VO_OK;
}

// ---- transform ------------------------------------------------------------------------------
int vo_transform_points_dev(vo_ctx* c, const float T[16], const float* d_T16, const float* d_in, int n,
                            const int* d_n, float* d_out) {
  VO_REQUIRE(c && (T || d_T16), "null argument");
  VO_REQUIRE(n >= 0 && (n == 0 || (d_in && d_out)), "bad point arrays");
  if (int r = set_device(c)) return r;
  if (d_T16) VO_HIP_CHECK(launch_transform_points_devpose(c->stream, d_T16, d_in, n, d_n, d_out));
  else VO_HIP_CHECK(launch_transform_points(c->stream, pose_from_T16(T), d_in, n, d_n, d_out));
  return VO_OK;
}

int vo_transform_points(vo_ctx* c, const float T[16], const float* in, int n, float* out) {
  VO_REQUIRE(c && T, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n >= 0 && (n == 0 || (in && out)), "bad point arrays");
  if (n == 0) return VO_OK;
  if (int r = set_device(c)) return r;
  if (int r = upload(c, c->in[0], in, sizeof(float) * 3 * (size_t)n)) return r;
  VO_HIP_CHECK(c->out[0].ensure(sizeof(float) * 3 * (size_t)n, c->stream));
  if (int r = vo_transform_points_dev(c, T, nullptr, c->in[0].as<float>(), n, nullptr, c->out[0].as<float>())) return r;
  VO_HIP_CHECK(hipMemcpyAsync(out, c->out[0].p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  return VO_OK;
}

// ---- triangulation ----------------------------------------------------------------------------
int vo_triangulate_dev(vo_ctx* c, const float K[9], const float X[16], const float* d_X16,
                       const int32_t* d_pairs, int n, const int* d_n, const float* d_p1, int n1,
                       const float* d_p2, int n2, const float* d_app2, float* d_out_xyz,
                       int32_t* d_out_pairs, float* d_out_app, int* d_n_out) {
  VO_REQUIRE(c && K && d_n_out, "null argument");
  VO_REQUIRE(X || d_X16, "no pose given");
  VO_REQUIRE(n >= 0 && n1 >= 0 && n2 >= 0, "negative count");
  VO_REQUIRE(n == 0 || (d_pairs && d_p1 && d_p2 && d_out_xyz), "null device array");
  VO_REQUIRE(aligned8(d_pairs, d_p1, d_p2, d_app2, d_out_app) && aligned8(d_out_pairs), "device array not on an 8-byte boundary");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(c->scratch.ensure(triangulate_scratch_bytes(n, 1), c->stream));
  Pose Xp;
  if (X) Xp = pose_from_T16(X);
  VO_HIP_CHECK(launch_triangulate(c->stream, K, X ? &Xp : nullptr, d_X16, d_pairs, n, d_n, d_p1, n1, d_p2,
                                  n2, d_app2, d_out_xyz, d_out_pairs, d_out_app, d_n_out,
                                  c->scratch.as<int>()));
  return VO_OK;
}

int vo_triangulate(vo_ctx* c, const float K[9], const float X[16], const int32_t* pairs, int n,
                   const float* p1, int n1, const float* p2, int n2, const float* app2, float* out_xyz,
                   int32_t* out_pairs, float* out_app, int* n_out) {
  VO_REQUIRE(c && K && X && n_out, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n >= 0 && n1 >= 0 && n2 >= 0, "negative count");
  VO_REQUIRE(n == 0 || (pairs && p1 && p2 && out_xyz), "null array");
  if (int r = set_device(c)) return r;
  const bool want_app = app2 && out_app;
  if (int r = upload(c, c->in[0], pairs, sizeof(int32_t) * 2 * (size_t)n)) return r;
  if (int r = upload(c, c->in[1], p1, sizeof(float) * 2 * (size_t)n1)) return r;
  if (int r = upload(c, c->in[2], p2, sizeof(float) * 2 * (size_t)n2)) return r;
  if (want_app) if (int r = upload(c, c->in[3], app2, sizeof(float) * 10 * (size_t)n2)) return r;
  const size_t nn = (size_t)(n ? n : 1);
  VO_HIP_CHECK(c->out[0].ensure(sizeof(float) * 3 * nn, c->stream));
  VO_HIP_CHECK(c->out[1].ensure(sizeof(int32_t) * 2 * nn, c->stream));
  if (want_app) VO_HIP_CHECK(c->out[2].ensure(sizeof(float) * 10 * nn, c->stream));
  if (int r = ensure_counts(c)) return r;
  if (int r = vo_triangulate_dev(c, K, X, nullptr, c->in[0].as<int32_t>(), n, nullptr, c->in[1].as<float>(),
                                 n1, c->in[2].as<float>(), n2, want_app ? c->in[3].as<float>() : nullptr,
                                 c->out[0].as<float>(), out_pairs ? c->out[1].as<int32_t>() : nullptr,
                                 want_app ? c->out[2].as<float>() : nullptr, c->counts.as<int>()))
    return r;
  int h = 0;
  VO_HIP_CHECK(hipMemcpyAsync(&h, c->counts.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (h > 0) {
    VO_HIP_CHECK(hipMemcpyAsync(out_xyz, c->out[0].p, sizeof(float) * 3 * (size_t)h, hipMemcpyDeviceToHost, c->stream));
    if (out_pairs) VO_HIP_CHECK(hipMemcpyAsync(out_pairs, c->out[1].p, sizeof(int32_t) * 2 * (size_t)h, hipMemcpyDeviceToHost, c->stream));
    if (want_app) VO_HIP_CHECK(hipMemcpyAsync(out_app, c->out[2].p, sizeof(float) * 10 * (size_t)h, hipMemcpyDeviceToHost, c->stream));
    VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  *n_out = h;
  return VO_OK;
}

// ---- epipolar initialisation (epipolar_utils.cpp:176-213) --------------------------------------
// Device half first (epi.hip): maxima of both images, the 45 sums of A^T A, one small read-back; then the host's 9 x 9
// eigen-solve, rank-2 projection and decomposition of E in double (include/vo/epipolar.hpp); then ONE launch that counts,
// for all four candidates at once, the correspondences that triangulate in front of both cameras; a second read-back.
int vo_estimate_transform_dev(vo_ctx* c, const float K[9], const int32_t* d_pairs, int n_max, const int* d_n, const float* d_p1,
                              int n1, const float* d_p2, int n2, float X_out[16]) {
  VO_REQUIRE(c && K && d_pairs && d_p1 && d_p2 && X_out, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n_max >= 8, "fewer than 8 correspondences");
  VO_REQUIRE(n1 > 0 && n2 > 0, "empty point set");
  VO_REQUIRE(aligned8(d_pairs, d_p1, d_p2), "device arrays must be 8-byte aligned");
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(c->epi_ws.ensure(epi_workspace_bytes(), c->stream));
  VO_HIP_CHECK(launch_epi_front(c->stream, d_pairs, n_max, d_n, d_p1, n1, d_p2, n2, c->epi_ws.p));
  struct { unsigned maxima[4]; int info[4]; int votes[8]; double ata[45]; } h;
  static_assert(sizeof(h) == 64 + 45 * sizeof(double), "layout of the epipolar workspace");
  VO_HIP_CHECK(hipMemcpyAsync(&h, c->epi_ws.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (h.info[1] > 0)
    return fail(VO_ERR_BAD_INDEX, "vo_estimate_transform: %d pair(s) index outside the point arrays", h.info[1]);
  if (h.info[0] < 8) return fail(VO_ERR_INVALID_ARG, "vo_estimate_transform: fewer than 8 correspondences (%d)", h.info[0]);
  std::vector<double> AtA(81, 0.0);
  for (int i = 0, k = 0; i < 9; ++i)
    for (int j = i; j < 9; ++j, ++k) AtA[(size_t)i * 9 + j] = AtA[(size_t)j * 9 + i] = h.ata[k];
  float mx[4];
  memcpy(mx, h.maxima, sizeof(mx));
  vo::Matrix3f k;
  for (int j = 0; j < 9; ++j) k.m[j] = K[j];
  const vo::Matrix3f F = vo::fundamental_from_normal_matrix(AtA, vo::conditioning_matrix(mx[0], mx[1]), vo::conditioning_matrix(mx[2], mx[3]));
  vo::Isometry3f X[4];
  vo::transform_candidates(k, F, X);
  Pose P[4];
  for (int q = 0; q < 4; ++q) P[q] = pose_from_T16(X[q].data());
  VO_HIP_CHECK(launch_epi_vote(c->stream, K, P, d_pairs, n_max, d_n, d_p1, n1, d_p2, n2, c->epi_ws.p));
  int votes[4] = {0, 0, 0, 0};
  VO_HIP_CHECK(hipMemcpyAsync(votes, static_cast<char*>(c->epi_ws.p) + 32, sizeof(votes), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  const vo::Isometry3f best = vo::pick_candidate(X, votes);
  memcpy(X_out, best.data(), sizeof(float) * 16);
  return VO_OK;
}

int vo_estimate_transform(vo_ctx* c, const float K[9], const int32_t* pairs, int n, const float* p1, int n1,
                          const float* p2, int n2, float X_out[16]) {
  VO_REQUIRE(c && K && pairs && p1 && p2 && X_out, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(n >= 8, "fewer than 8 correspondences");
  VO_REQUIRE(n1 > 0 && n2 > 0, "empty point set");
  for (int i = 0; i < n; ++i)
    if (pairs[2 * i] < 0 || pairs[2 * i] >= n1 || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= n2)
      return fail(VO_ERR_BAD_INDEX, "vo_estimate_transform: pair %d = (%d,%d) outside the point arrays", i,
                  pairs[2 * i], pairs[2 * i + 1]);
  if (int r = set_device(c)) return r;
  if (int r = upload(c, c->in[0], pairs, sizeof(int32_t) * 2 * (size_t)n)) return r;
  if (int r = upload(c, c->in[1], p1, sizeof(float) * 2 * (size_t)n1)) return r;
  if (int r = upload(c, c->in[2], p2, sizeof(float) * 2 * (size_t)n2)) return r;
  return vo_estimate_transform_dev(c, K, c->in[0].as<int32_t>(), n, nullptr, c->in[1].as<float>(), n1, c->in[2].as<float>(), n2, X_out);
}

// ---- the map (PointCloudVector<3>::update + the history chain of vo_complete.cpp:145-147,175-176,183) ---------------------------
struct vo_map {
  vo_ctx* ctx = nullptr;
  unsigned long long ctx_id = 0;
  MapDev d;
  DevBuf scratch;
  float* hist = nullptr;        // [0,16) the history isometry, [16,32) staging of a host isometry (vo_map_update)
  long long size_ub = 0;        // upper bound of the size known without asking the device
  DevBuf up_xyz, up_app;        // staging of vo_map_update
};

constexpr int MAP_MAX_ENTRIES = 1 << 29;      // 2.5 x that many table slots still fit 32 bits
static unsigned map_tcap_for(int cap) {
  unsigned t = 1024;
  while (t < 2u * (unsigned)cap + 2u * ((unsigned)cap >> 2)) t <<= 1;      // at most 40 % full (cap <= MAP_MAX_ENTRIES: t <= 2^31)
  return t;
}

static void map_free_arrays(MapDev& d) {
  if (d.pts) (void)hipFree(d.pts);
  if (d.app) (void)hipFree(d.app);
  if (d.table) (void)hipFree(d.table);
  if (d.last) (void)hipFree(d.last);
  d.pts = d.app = nullptr; d.table = nullptr; d.last = nullptr;
}

static int map_alloc_arrays(MapDev& d, int cap) {
  d.cap = cap; d.tcap = map_tcap_for(cap);
  VO_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d.pts), sizeof(float) * 3 * (size_t)cap));
  VO_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d.app), sizeof(float) * 10 * (size_t)cap));
  VO_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d.table), sizeof(unsigned long long) * (size_t)d.tcap));
  VO_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d.last), sizeof(int) * (size_t)d.tcap));
  return VO_OK;
}

// room for n more entries: the bound the host keeps (size_ub grows by a cloud's row count per update) is refreshed from
// the device only when it would pass the capacity, and the arrays grow only when the true size would
static int map_reserve(vo_map* m, int n) {
  vo_ctx* c = m->ctx;
  if (m->size_ub + n <= m->d.cap) return VO_OK;
  if (c->capturing) return fail(VO_ERR_NOT_READY, "vo_map: the map would have to grow inside a graph capture (create it with the capacity it will need)");
  int size = 0;
  VO_HIP_CHECK(hipMemcpyAsync(&size, m->d.hdr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  m->size_ub = size;
  if ((long long)size + n <= m->d.cap) return VO_OK;
  long long want = 2ll * m->d.cap;
  if (want < (long long)size + n) want = (long long)size + n + ((long long)size + n) / 2;
  if ((long long)size + n > MAP_MAX_ENTRIES) return fail(VO_ERR_INVALID_ARG, "vo_map: more than 2^29 entries");
  if (want > MAP_MAX_ENTRIES) want = MAP_MAX_ENTRIES;
  MapDev old = m->d;
  MapDev nd = old;
  nd.pts = nd.app = nullptr; nd.table = nullptr; nd.last = nullptr;
  if (int r = map_alloc_arrays(nd, (int)want)) { map_free_arrays(nd); return r; }
  if (size > 0) {
    VO_HIP_CHECK(hipMemcpyAsync(nd.pts, old.pts, sizeof(float) * 3 * (size_t)size, hipMemcpyDeviceToDevice, c->stream));
    VO_HIP_CHECK(hipMemcpyAsync(nd.app, old.app, sizeof(float) * 10 * (size_t)size, hipMemcpyDeviceToDevice, c->stream));
  }
  VO_HIP_CHECK(launch_map_rehash(c->stream, nd, size));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  map_free_arrays(old);
  m->d = nd;
  return VO_OK;
}

#define VO_MAP_LIVE(m) \
  do { VO_REQUIRE(m, "null map"); VO_REQUIRE(ctx_alive((m)->ctx, (m)->ctx_id), "the context this map was made on has been destroyed"); } while (0)

int vo_map_create(vo_ctx* c, int capacity, vo_map** out) {
  VO_REQUIRE(c && out, "null argument");
  VO_NOT_CAPTURING(c);
  VO_REQUIRE(capacity >= 0 && capacity <= MAP_MAX_ENTRIES, "bad capacity (at most 2^29 entries)");
  *out = nullptr;
  if (int r = set_device(c)) return r;
  vo_map* m = new vo_map();
  m->ctx = c; m->ctx_id = c->id;
  const int cap = capacity < 1024 ? 1024 : capacity;
  int rc = map_alloc_arrays(m->d, cap);
  if (rc == VO_OK && hipMalloc(reinterpret_cast<void**>(&m->d.hdr), 64) != hipSuccess) rc = fail(VO_ERR_OUT_OF_MEMORY, "vo_map_create: out of device memory");
  if (rc == VO_OK && hipMalloc(reinterpret_cast<void**>(&m->hist), sizeof(float) * 32) != hipSuccess) rc = fail(VO_ERR_OUT_OF_MEMORY, "vo_map_create: out of device memory");
  if (rc != VO_OK) { (void)hipGetLastError(); map_free_arrays(m->d); if (m->d.hdr) (void)hipFree(m->d.hdr); if (m->hist) (void)hipFree(m->hist); delete m; return rc; }
  *out = m;
  return vo_map_clear(m);
}

int vo_map_destroy(vo_map* m) {
  if (!m) return VO_OK;
  if (ctx_alive(m->ctx, m->ctx_id)) {
    VO_NOT_CAPTURING(m->ctx);
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
  } else {
    (void)hipDeviceSynchronize();
  }
  map_free_arrays(m->d);
  if (m->d.hdr) (void)hipFree(m->d.hdr);
  if (m->hist) (void)hipFree(m->hist);
  m->scratch.release(); m->up_xyz.release(); m->up_app.release();
  delete m;
  return VO_OK;
}

int vo_map_clear(vo_map* m) {
  VO_MAP_LIVE(m);
  vo_ctx* c = m->ctx;
  VO_NOT_CAPTURING(c);
  if (int r = set_device(c)) return r;
  VO_HIP_CHECK(hipMemsetAsync(m->d.hdr, 0, 64, c->stream));
  VO_HIP_CHECK(launch_map_rehash(c->stream, m->d, 0));
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  VO_HIP_CHECK(hipMemcpyAsync(m->hist, I, sizeof(I), hipMemcpyHostToDevice, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));
  m->size_ub = 0;
  return VO_OK;
}

int vo_map_update_dev(vo_map* m, const float* d_xyz, const float* d_app, int n_max, const int* d_n, const float* d_T16) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(n_max >= 0 && n_max <= MAP_MAX_ENTRIES && (n_max == 0 || (d_xyz && d_app)), "bad cloud");
  VO_REQUIRE(aligned8(d_app), "device appearance rows must be 8-byte aligned");
  if (n_max == 0) return VO_OK;
  vo_ctx* c = m->ctx;
  if (int r = set_device(c)) return r;
  if (int r = map_reserve(m, n_max)) return r;
  VO_HIP_CHECK(m->scratch.ensure(sizeof(int) * map_scratch_ints(n_max), c->stream));
  VO_HIP_CHECK(launch_map_update(c->stream, m->d, d_xyz, d_app, n_max, d_n, d_T16, m->scratch.as<int>()));
  m->size_ub += n_max;
  return VO_OK;
}

int vo_map_update(vo_map* m, const float* xyz, const float* app, int n, const float T16[16]) {
  VO_MAP_LIVE(m);
  VO_NOT_CAPTURING(m->ctx);
  VO_REQUIRE(n >= 0 && (n == 0 || (xyz && app)), "bad cloud");
  if (n == 0) return VO_OK;
  vo_ctx* c = m->ctx;
  if (int r = set_device(c)) return r;
  if (int r = upload(c, m->up_xyz, xyz, sizeof(float) * 3 * (size_t)n)) return r;
  if (int r = upload(c, m->up_app, app, sizeof(float) * 10 * (size_t)n)) return r;
  if (T16) VO_HIP_CHECK(hipMemcpyAsync(m->hist + 16, T16, sizeof(float) * 16, hipMemcpyHostToDevice, c->stream));
  VO_HIP_CHECK(hipStreamSynchronize(c->stream));          // the host arrays may go away after the call
  return vo_map_update_dev(m, m->up_xyz.as<float>(), m->up_app.as<float>(), n, nullptr, T16 ? m->hist + 16 : nullptr);
}

int vo_map_history_reset_dev(vo_map* m, const float* d_X16) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(d_X16, "null pose");
  if (int r = set_device(m->ctx)) return r;
  VO_HIP_CHECK(launch_map_history(m->ctx->stream, m->hist, d_X16, 1));
  return VO_OK;
}

int vo_map_history_step_dev(vo_map* m, const float* d_X16) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(d_X16, "null pose");
  if (int r = set_device(m->ctx)) return r;
  VO_HIP_CHECK(launch_map_history(m->ctx->stream, m->hist, d_X16, 0));
  return VO_OK;
}

int vo_map_history_dev_ptr(vo_map* m, const float** d_T16) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(d_T16, "null argument");
  *d_T16 = m->hist;
  return VO_OK;
}

int vo_map_get_history(vo_map* m, float T16[16]) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(T16, "null argument");
  VO_NOT_CAPTURING(m->ctx);
  if (int r = set_device(m->ctx)) return r;
  VO_HIP_CHECK(hipMemcpyAsync(T16, m->hist, sizeof(float) * 16, hipMemcpyDeviceToHost, m->ctx->stream));
  VO_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return VO_OK;
}

int vo_map_size(vo_map* m, int* n) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(n, "null argument");
  VO_NOT_CAPTURING(m->ctx);
  if (int r = set_device(m->ctx)) return r;
  int h[4] = {0, 0, 0, 0};
  VO_HIP_CHECK(hipMemcpyAsync(h, m->d.hdr, sizeof(h), hipMemcpyDeviceToHost, m->ctx->stream));
  VO_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  m->size_ub = h[0];
  *n = h[0];
  if (h[3] > 0) return fail(VO_ERR_BAD_INDEX, "vo_map: %d entries were dropped for lack of room", h[3]);
  return VO_OK;
}

int vo_map_read(vo_map* m, float* xyz, float* app, int capacity, int* n_out) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(capacity >= 0 && n_out, "bad arguments");
  int n = 0;
  if (int r = vo_map_size(m, &n)) return r;
  *n_out = n;
  const int k = n < capacity ? n : capacity;
  if (k > 0 && xyz) VO_HIP_CHECK(hipMemcpyAsync(xyz, m->d.pts, sizeof(float) * 3 * (size_t)k, hipMemcpyDeviceToHost, m->ctx->stream));
  if (k > 0 && app) VO_HIP_CHECK(hipMemcpyAsync(app, m->d.app, sizeof(float) * 10 * (size_t)k, hipMemcpyDeviceToHost, m->ctx->stream));
  VO_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return VO_OK;
}

int vo_map_transform(vo_map* m, const float T16[16]) {
  VO_MAP_LIVE(m);
  VO_REQUIRE(T16, "null argument");
  if (int r = set_device(m->ctx)) return r;
  const long long bound = m->size_ub < m->d.cap ? m->size_ub : m->d.cap;
  VO_HIP_CHECK(launch_map_transform(m->ctx->stream, m->d, pose_from_T16(T16), (int)bound));
  return VO_OK;
}

int vo_map_dev_ptrs(vo_map* m, const float** d_xyz, const float** d_app, const int** d_size) {
  VO_MAP_LIVE(m);
  if (d_xyz) *d_xyz = m->d.pts;
  if (d_app) *d_app = m->d.app;
  if (d_size) *d_size = m->d.hdr;
  return VO_OK;
}

}  // extern "C"
