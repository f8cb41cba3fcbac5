// Step-level entry points of libmotioned.so: the launch list of one denoising step, recorded once and re-issued from C.
//
// SURVEY section 8(b) sketched `me_plan` ("sizes the activation arena, builds the launch list") and `me_denoise_step`.  The launch
// GRAPH of the step (which kernel on which rows, models/graph.py) stays host logic in Python; what moves behind the C ABI is its
// EXECUTION: while a plan records, every kernel the library launches on this thread (me_launch, me_common.h) is appended to it together
// with the cross-stream dependencies the host states through me_plan_event_record / me_plan_event_wait, and me_denoise_step() then
// replays the whole step -- ~1100 launches on two live HIP streams -- from one C call: no Python, no ctypes marshalling, no argument
// validation, no kernel selection per launch.  Unlike a captured hipGraph the replay keeps the two streams live (the two-stream
// fork / join inside a captured graph measured 1.2 % slower than live streams).
//
// Replaces (reference): the per-op Python dispatch under pipeline_motion_editor.py:603-648 (one loop body = one me_denoise_step).
#include "me_common.h"
#include "../../include/motioned.h"
#include <stdio.h>
#include <string.h>
#include <vector>

extern "C" void me_set_error(const char* msg);
extern "C" void me_set_hip_error(const char* what, int err);

namespace {

enum NodeKind : int { NODE_LAUNCH = 0, NODE_RECORD = 1, NODE_WAIT = 2 };

struct Node {
  int kind;
  int stream;          // index into me_plan::streams (0 = the plan's main stream)
  int event;           // NODE_RECORD / NODE_WAIT: index into me_plan::events
  const void* fn;      // NODE_LAUNCH: host-side kernel handle (what hipLaunchKernel takes)
  dim3 grid, block;
  unsigned lds;
  unsigned first_arg;  // index of the launch's first argument in me_plan::arg_off
  unsigned n_args;
};

}  // namespace

struct me_plan {
  std::vector<Node> nodes;
  std::vector<unsigned char> blob;     // argument bytes of every launch, each argument 16-byte aligned
  std::vector<size_t> arg_off;         // per argument: offset into blob
  std::vector<unsigned> arg_size;      // per argument: bytes
  std::vector<void*> argv;             // per argument: &blob[arg_off] (built by me_plan_end, the blob no longer moves)
  std::vector<hipStream_t> streams;    // [0] = main (the stream given to me_plan_begin; me_denoise_step substitutes its own)
  std::vector<hipEvent_t> events;      // created by me_plan_end, one per recorded event
  int n_events = 0;
  bool recording = false, ended = false;
  long n_launch = 0, n_record = 0, n_wait = 0;
  // the static buffers the recorded step reads and writes (me_plan_bind)
  void* lat_in = nullptr;
  void* lat_out = nullptr;
  void* text = nullptr;
  float* params = nullptr;
  int64_t lat_bytes = 0, text_bytes = 0;
  long replays = 0;
};

namespace {

thread_local me_plan* g_rec = nullptr;

int stream_index(me_plan* p, hipStream_t s) {
  for (size_t i = 0; i < p->streams.size(); ++i)
    if (p->streams[i] == s) return (int)i;
  p->streams.push_back(s);
  return (int)p->streams.size() - 1;
}

__global__ void plan_params_kernel(float* p, float t, float guidance, float ca, float cb) {
  if (threadIdx.x == 0) {
    p[0] = t;
    p[1] = guidance;
    p[2] = ca;
    p[3] = cb;
  }
}

}  // namespace

extern "C" int me_plan_recording(void) { return g_rec != nullptr; }

extern "C" void me_plan_append_launch(const void* fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned lds_bytes, void* stream,
                                      void* const* args, const unsigned* arg_bytes, int n_args) {
  me_plan* p = g_rec;
  if (!p) return;
  Node n{};
  n.kind = NODE_LAUNCH;
  n.stream = stream_index(p, reinterpret_cast<hipStream_t>(stream));
  n.event = -1;
  n.fn = fn;
  n.grid = dim3(gx, gy, gz);
  n.block = dim3(bx, by, bz);
  n.lds = lds_bytes;
  n.first_arg = (unsigned)p->arg_off.size();
  n.n_args = (unsigned)n_args;
  for (int i = 0; i < n_args; ++i) {
    const size_t off = (p->blob.size() + 15) & ~(size_t)15;
    p->blob.resize(off + arg_bytes[i]);
    memcpy(p->blob.data() + off, args[i], arg_bytes[i]);
    p->arg_off.push_back(off);
    p->arg_size.push_back(arg_bytes[i]);
  }
  p->nodes.push_back(n);
  ++p->n_launch;
}

extern "C" int me_plan_begin(me_plan** out, void* main_stream) {
  if (!out) { me_set_error("me_plan_begin: out is NULL"); return ME_EINVAL; }
  if (g_rec) { me_set_error("me_plan_begin: this thread is already recording a plan"); return ME_EINVAL; }
  me_plan* p = new me_plan();
  p->streams.push_back(reinterpret_cast<hipStream_t>(main_stream));
  p->recording = true;
  g_rec = p;
  *out = p;
  return ME_OK;
}

extern "C" int me_plan_event_record(void* stream, int32_t* event_id) {
  me_plan* p = g_rec;
  if (!p) { me_set_error("me_plan_event_record: no plan is recording on this thread"); return ME_EINVAL; }
  if (!event_id) { me_set_error("me_plan_event_record: event_id is NULL"); return ME_EINVAL; }
  Node n{};
  n.kind = NODE_RECORD;
  n.stream = stream_index(p, reinterpret_cast<hipStream_t>(stream));
  n.event = p->n_events++;
  p->nodes.push_back(n);
  ++p->n_record;
  *event_id = n.event;
  return ME_OK;
}

extern "C" int me_plan_event_wait(void* stream, int32_t event_id) {
  me_plan* p = g_rec;
  if (!p) { me_set_error("me_plan_event_wait: no plan is recording on this thread"); return ME_EINVAL; }
  if (event_id < 0 || event_id >= p->n_events) { me_set_error("me_plan_event_wait: unknown event id (record it first, inside the same plan)"); return ME_EINVAL; }
  Node n{};
  n.kind = NODE_WAIT;
  n.stream = stream_index(p, reinterpret_cast<hipStream_t>(stream));
  n.event = event_id;
  p->nodes.push_back(n);
  ++p->n_wait;
  return ME_OK;
}

extern "C" int me_plan_end(me_plan* p) {
  if (!p || p != g_rec) { me_set_error("me_plan_end: not the plan this thread is recording"); return ME_EINVAL; }
  g_rec = nullptr;
  p->recording = false;
  p->argv.resize(p->arg_off.size());
  for (size_t i = 0; i < p->arg_off.size(); ++i) p->argv[i] = p->blob.data() + p->arg_off[i];
  p->events.assign((size_t)p->n_events, nullptr);
  for (int i = 0; i < p->n_events; ++i) {
    const hipError_t e = hipEventCreateWithFlags(&p->events[(size_t)i], hipEventDisableTiming);
    if (e != hipSuccess) {
      me_set_hip_error("me_plan_end: hipEventCreateWithFlags", (int)e);
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(p->events[(size_t)j]);
      p->events.clear();
      p->n_events = 0;
      return ME_EHIP;   // the launch list itself stays readable (me_plan_info / me_plan_node); me_denoise_step refuses the plan
    }
  }
  p->ended = true;
  return ME_OK;
}

extern "C" int me_plan_bind(me_plan* p, void* latents_in, int64_t latents_bytes, void* text_emb, int64_t text_bytes, float* step_params, void* latents_out) {
  if (!p || p->recording) { me_set_error("me_plan_bind: NULL plan, or still recording"); return ME_EINVAL; }
  if (!latents_in || !latents_out || !step_params || latents_bytes <= 0 || text_bytes < 0 || (text_bytes > 0 && !text_emb)) {
    me_set_error("me_plan_bind: bad arguments");
    return ME_EINVAL;
  }
  p->lat_in = latents_in;
  p->lat_out = latents_out;
  p->text = text_emb;
  p->params = step_params;
  p->lat_bytes = latents_bytes;
  p->text_bytes = text_bytes;
  return ME_OK;
}

extern "C" int me_plan_info(const me_plan* p, me_plan_stats* out) {
  if (!p || !out) { me_set_error("me_plan_info: NULL argument"); return ME_EINVAL; }
  out->launches = p->n_launch;
  out->event_records = p->n_record;
  out->event_waits = p->n_wait;
  out->streams = (int32_t)p->streams.size();
  out->arg_bytes = (int64_t)p->blob.size();
  out->replays = p->replays;
  return ME_OK;
}

extern "C" int me_plan_node(const me_plan* p, int64_t index, me_plan_node_info* out, void* arg_bytes_out, int64_t arg_bytes_cap) {
  if (!p || !out || index < 0 || index >= (int64_t)p->nodes.size()) { me_set_error("me_plan_node: bad arguments"); return ME_EINVAL; }
  const Node& n = p->nodes[(size_t)index];
  out->kind = n.kind;
  out->stream = n.stream;
  out->event = n.event;
  out->grid[0] = n.grid.x, out->grid[1] = n.grid.y, out->grid[2] = n.grid.z;
  out->block[0] = n.block.x, out->block[1] = n.block.y, out->block[2] = n.block.z;
  out->lds_bytes = n.lds;
  out->n_args = (int32_t)n.n_args;
  // the arguments, densely packed in declaration order (each at its own size): what a test compares with the values it passed
  int64_t total = 0;
  for (unsigned i = 0; i < n.n_args; ++i) {
    const size_t off = p->arg_off[n.first_arg + i], sz = p->arg_size[n.first_arg + i];
    if (arg_bytes_out && total + (int64_t)sz <= arg_bytes_cap) memcpy(static_cast<unsigned char*>(arg_bytes_out) + total, p->blob.data() + off, sz);
    total += (int64_t)sz;
  }
  out->arg_bytes = total;
  return ME_OK;
}

extern "C" int me_denoise_step(me_plan* p, const void* latents_in, void* latents_out, const void* text_emb, float t, float guidance, float ca, float cb, void* stream) {
  if (!p || !p->ended) { me_set_error("me_denoise_step: NULL plan, or one that me_plan_end has not completed"); return ME_EINVAL; }
  if (g_rec) { me_set_error("me_denoise_step: a plan is recording on this thread"); return ME_EINVAL; }
  if (!p->lat_in || !p->params) { me_set_error("me_denoise_step: the plan's buffers are not bound (me_plan_bind)"); return ME_EINVAL; }
  if (text_emb && !p->text) { me_set_error("me_denoise_step: text_emb given but the plan has no bound text buffer"); return ME_EINVAL; }
  hipStream_t main = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  hipError_t e = hipSuccess;
  long at = -1;
#define ME_PLAN_TRY(call, where) do { if (e == hipSuccess) { e = (call); if (e != hipSuccess) at = (where); } } while (0)
  if (latents_in && latents_in != p->lat_in) ME_PLAN_TRY(hipMemcpyAsync(p->lat_in, latents_in, (size_t)p->lat_bytes, hipMemcpyDeviceToDevice, main), -2);
  if (text_emb && text_emb != p->text) ME_PLAN_TRY(hipMemcpyAsync(p->text, text_emb, (size_t)p->text_bytes, hipMemcpyDeviceToDevice, main), -3);
  if (e == hipSuccess) {
    void* pa[5];
    float* pp = p->params;
    pa[0] = &pp, pa[1] = &t, pa[2] = &guidance, pa[3] = &ca, pa[4] = &cb;
    ME_PLAN_TRY(hipLaunchKernel(reinterpret_cast<const void*>(plan_params_kernel), dim3(1), dim3(64), pa, 0, main), -4);
  }
  const size_t nn = p->nodes.size();
  for (size_t i = 0; i < nn && e == hipSuccess; ++i) {
    const Node& n = p->nodes[i];
    hipStream_t s = n.stream == 0 ? main : p->streams[(size_t)n.stream];
    switch (n.kind) {
      case NODE_LAUNCH: e = hipLaunchKernel(n.fn, n.grid, n.block, n.n_args ? &p->argv[n.first_arg] : nullptr, n.lds, s); break;
      case NODE_RECORD: e = hipEventRecord(p->events[(size_t)n.event], s); break;
      default: e = hipStreamWaitEvent(s, p->events[(size_t)n.event], 0); break;
    }
    if (e != hipSuccess) at = (long)i;
  }
  if (latents_out && latents_out != p->lat_out) ME_PLAN_TRY(hipMemcpyAsync(latents_out, p->lat_out, (size_t)p->lat_bytes, hipMemcpyDeviceToDevice, main), -5);
#undef ME_PLAN_TRY
  if (e != hipSuccess) {
    char what[96];
    snprintf(what, sizeof(what), "me_denoise_step: node %ld", at);
    me_set_hip_error(what, (int)e);
    return ME_EHIP;
  }
  ++p->replays;
  return ME_OK;
}

extern "C" void me_plan_destroy(me_plan* p) {
  if (!p) return;
  if (p == g_rec) g_rec = nullptr;
  for (hipEvent_t ev : p->events)
    if (ev) (void)hipEventDestroy(ev);
  delete p;
}
