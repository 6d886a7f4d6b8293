// tfimm_hip_plan_*: a whole forward behind three C calls (SURVEY.md §8b sketch: program_create / forward).
//
// The op-level entry points of include/tfimm_hip.h are what a TF-free tfimm binds; turning a model configuration into
// the sequence of such calls (lowering, weight packing, buffer planning, tile selection) is host logic that lives in
// Python (tfimm/engine/graph.py).  A host WITHOUT Python does not have to re-implement it: graph.Plan.export() writes the
// finished plan of one (model, batch size) -- every call with its arguments, the packed constants, the slab sizes -- into
// a self-contained blob, and the functions below execute it:
//
//     tfimm_hip_plan_query(blob, bytes, &info)                     sizes: device workspace, input, outputs
//     tfimm_hip_plan_create(blob, bytes, workspace, stream, &plan)  uploads the constants into the caller's workspace,
//                                                                   resolves every pointer of the call list
//     tfimm_hip_plan_forward(plan, input, in_dtype, stream)         enqueues the forward (asynchronous, capturable)
//     tfimm_hip_plan_output(plan, "logits", &ptr, &rows, &cols, &dtype)   where the result lies in the workspace
//     tfimm_hip_plan_destroy(plan)
//
// Same conventions as the rest of the ABI: plain C, the caller owns the device memory (one workspace allocation), every
// launch is stream-ordered, nothing allocates on the device.  The executor adds no arithmetic: it calls the very entry
// points a Python plan calls, with the same arguments, so the result is bit-identical.
#include "common.h"

#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace {

// ---- blob layout (little endian; written by tfimm/engine/graph.py Plan.export) ------------------------------------------
constexpr uint32_t kMagic = 0x4c504654u;   // "TFPL"
constexpr uint32_t kVersion = 1;
enum ArgKind : uint32_t { A_INT = 0, A_FLOAT = 1, A_NULL = 2, A_SLAB = 3, A_CONST = 4, A_HOST = 5, A_INPUT = 6, A_STRUCT = 7 };

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  template <typename T> T get() {
    T v{};
    if (p + sizeof(T) > end) { ok = false; return v; }
    memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  std::string str() {
    const uint32_t n = get<uint32_t>();
    if (!ok || p + n > end) { ok = false; return std::string(); }
    std::string s(reinterpret_cast<const char*>(p), n);
    p += n;
    return s;
  }
};

struct ArgSpec { uint32_t kind, aux; uint64_t value; };
struct Arg { int64_t i; double f; void* p; };
struct Reloc { uint32_t field_offset, kind, aux; uint64_t value; };
struct StructSpec { std::vector<uint8_t> bytes; std::vector<Reloc> relocs; };
struct Output { std::string name; uint32_t slab; uint64_t offset; uint64_t rows_per_image, cols; uint32_t dtype; };

typedef int (*adapter_fn)(const Arg*, void*);
struct Call { adapter_fn fn; std::string name; std::vector<ArgSpec> spec; std::vector<Arg> args; bool is_memset = false; };

// ---- generic adapter: Arg[] -> the typed parameters of an entry point (the last parameter is always the stream) ----------
template <typename T> T as(const Arg& a) {
  if constexpr (std::is_pointer<T>::value) return reinterpret_cast<T>(a.p);
  else if constexpr (std::is_floating_point<T>::value) return static_cast<T>(a.f);
  else return static_cast<T>(a.i);
}
template <typename... P, size_t... I>
int invoke_impl(int (*fn)(P..., void*), const Arg* a, void* stream, std::index_sequence<I...>) {
  return fn(as<P>(a[I])..., stream);
}
template <typename R, typename... P> constexpr size_t arity(R (*)(P...)) { return sizeof...(P) - 1; }

// peel the trailing void* off the parameter pack: declare adapters through a macro that knows the full type
#define TFIMM_ADAPTER(NAME, ...)                                                                          \
  {#NAME, {[](const Arg* a, void* s) -> int {                                                             \
             return invoke_impl<__VA_ARGS__>(NAME, a, s, std::make_index_sequence<arity(NAME)>{});       \
           }, arity(NAME), 0}}
// entry points that take ONE descriptor: its size is recorded so that a blob's struct can be checked against it
#define TFIMM_ADAPTER_DESC(NAME, DESC)                                                                    \
  {#NAME, {[](const Arg* a, void* s) -> int {                                                             \
             return invoke_impl<const DESC*>(NAME, a, s, std::make_index_sequence<arity(NAME)>{});       \
           }, arity(NAME), sizeof(DESC)}}

struct Entry { adapter_fn fn; size_t nargs; size_t desc_bytes; };

const std::map<std::string, Entry>& table() {
  using vp = const void*;
  using mp = void*;
  using fp = const float*;
  static const std::map<std::string, Entry> t = {
      TFIMM_ADAPTER_DESC(tfimm_hip_gemm, tfimm_gemm_desc),
      TFIMM_ADAPTER_DESC(tfimm_hip_conv_chain, tfimm_chain_desc),
      TFIMM_ADAPTER_DESC(tfimm_hip_mlp_fused, tfimm_mlp_desc),
      TFIMM_ADAPTER_DESC(tfimm_hip_expand_dwconv, tfimm_expand_dw_desc),
      TFIMM_ADAPTER_DESC(tfimm_hip_stem_conv_pool, tfimm_stem_desc),
      TFIMM_ADAPTER_DESC(tfimm_hip_attention, tfimm_attn_desc),
      TFIMM_ADAPTER_DESC(tfimm_hip_talking_heads_attention, tfimm_tha_desc),
      TFIMM_ADAPTER(tfimm_hip_cast_input, vp, int, mp, int64_t, int, int),
      TFIMM_ADAPTER(tfimm_hip_cast_input_pad, vp, int, mp, int, int, int, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_row_stats, vp, float*, int64_t, int, int64_t, float),
      TFIMM_ADAPTER(tfimm_hip_layernorm, vp, mp, fp, fp, int64_t, int, int64_t, int64_t, float),
      TFIMM_ADAPTER(tfimm_hip_class_attention, vp, vp, mp, int, int, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_copy_rows, vp, mp, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_maxpool, vp, mp, int, int, int, int, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_mean_rows, vp, mp, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_bcast_rows, vp, mp, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_dwconv, vp, fp, fp, mp, mp, int, int, int, int, int, int, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_se_gate, vp, int, float, fp, fp, fp, fp, float*, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_scale_channels, vp, fp, vp, mp, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_patch_merge_ln, vp, mp, fp, fp, int, int, int, int, float),
      TFIMM_ADAPTER(tfimm_hip_attention_probs, vp, mp, int, int, int, int, float),
      TFIMM_ADAPTER(tfimm_hip_group_norm, vp, fp, fp, vp, mp, mp, int, int, int, int, float, int, int),
      TFIMM_ADAPTER(tfimm_hip_blur_pool, vp, mp, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_avg_pool, vp, mp, int, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_eca_gate, fp, float, fp, float*, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_grouped_conv3x3, vp, vp, fp, mp, int, int, int, int, int, int),
      TFIMM_ADAPTER(tfimm_hip_bias_act, vp, fp, mp, int64_t, int, int),
  };
  return t;
}

struct Plan {
  uint32_t batch = 0;
  std::vector<uint64_t> slab_bytes, slab_off;       // offsets inside the workspace
  std::vector<uint64_t> const_bytes, const_off, const_src;   // workspace offset / blob offset
  std::vector<std::vector<uint8_t>> host_consts;    // arrays entry points take by HOST pointer (talking-heads layers)
  std::vector<StructSpec> structs;
  std::vector<Call> calls;
  std::vector<Output> outputs;
  int32_t input_call = -1;                          // index of the input conversion call
  int32_t stem_call = -1, stem_struct = -1;         // fused ResNet stem that can read the caller's image directly
  int32_t img_h = 0, img_w = 0, img_c = 0, stem_pad_t = 0, stem_pad_l = 0;
  uint64_t workspace_bytes = 0;
  char* ws = nullptr;
};

uint64_t align256(uint64_t v) { return (v + 255) / 256 * 256; }

int parse(const void* blob, size_t bytes, Plan& pl) {
  Reader r{static_cast<const uint8_t*>(blob), static_cast<const uint8_t*>(blob) + bytes};
  if (r.get<uint32_t>() != kMagic) TFIMM_FAIL(TFIMM_EINVAL, "plan: not a plan blob");
  if (r.get<uint32_t>() != kVersion) TFIMM_FAIL(TFIMM_EINVAL, "plan: blob version mismatch");
  if ((int)r.get<uint32_t>() != tfimm_hip_abi_version()) TFIMM_FAIL(TFIMM_EINVAL, "plan: exported for another ABI version");
  pl.batch = r.get<uint32_t>();
  pl.img_h = r.get<int32_t>(); pl.img_w = r.get<int32_t>(); pl.img_c = r.get<int32_t>();
  // every slab / constant is bounded (2^40 bytes: far above the 288 GB of an MI355X) and there are at most 2^32 of each, so the
  // running workspace offset stays below 2^40 * 2^33 < 2^64 only if it is checked as well: a corrupted size near 2^64 would
  // otherwise wrap `off` to a SMALL workspace_bytes while resolve() keeps comparing offsets against the huge slab size
  constexpr uint64_t kMaxBuf = 1ull << 40;
  uint64_t off = 0;
  for (uint32_t i = 0, n = r.get<uint32_t>(); i < n && r.ok; ++i) {
    pl.slab_bytes.push_back(r.get<uint64_t>());
    pl.slab_off.push_back(off);
    if (pl.slab_bytes.back() > kMaxBuf || off > kMaxBuf) TFIMM_FAIL(TFIMM_EINVAL, "plan: slab %u of %llu bytes", i, (unsigned long long)pl.slab_bytes.back());
    off += align256(pl.slab_bytes.back());
  }
  for (uint32_t i = 0, n = r.get<uint32_t>(); i < n && r.ok; ++i) {
    pl.const_bytes.push_back(r.get<uint64_t>());
    pl.const_src.push_back(r.get<uint64_t>());
    pl.const_off.push_back(off);
    if (pl.const_bytes.back() > bytes || pl.const_src.back() > bytes - pl.const_bytes.back())      // (overflow-safe)
      TFIMM_FAIL(TFIMM_EINVAL, "plan: constant outside the blob");
    if (pl.const_bytes.back() > kMaxBuf || off > kMaxBuf) TFIMM_FAIL(TFIMM_EINVAL, "plan: workspace too large");
    off += align256(pl.const_bytes.back());
  }
  pl.workspace_bytes = off;
  for (uint32_t i = 0, n = r.get<uint32_t>(); i < n && r.ok; ++i) {
    const uint64_t nb = r.get<uint64_t>(), src = r.get<uint64_t>();
    if (nb > bytes || src > bytes - nb) TFIMM_FAIL(TFIMM_EINVAL, "plan: host constant outside the blob");
    pl.host_consts.emplace_back(static_cast<const uint8_t*>(blob) + src, static_cast<const uint8_t*>(blob) + src + nb);
  }
  for (uint32_t i = 0, n = r.get<uint32_t>(); i < n && r.ok; ++i) {
    StructSpec s;
    const uint32_t nb = r.get<uint32_t>(), nr = r.get<uint32_t>();
    if (!r.ok || nb > (size_t)(r.end - r.p)) TFIMM_FAIL(TFIMM_EINVAL, "plan: truncated struct");
    s.bytes.assign(r.p, r.p + nb);
    r.p += nb;
    for (uint32_t k = 0; k < nr && r.ok; ++k) {
      Reloc rl;
      rl.field_offset = r.get<uint32_t>(); rl.kind = r.get<uint32_t>(); rl.aux = r.get<uint32_t>(); rl.value = r.get<uint64_t>();
      if (nb < sizeof(void*) || rl.field_offset > nb - sizeof(void*)) TFIMM_FAIL(TFIMM_EINVAL, "plan: relocation outside its struct");
      s.relocs.push_back(rl);
    }
    pl.structs.push_back(std::move(s));
  }
  for (uint32_t i = 0, n = r.get<uint32_t>(); i < n && r.ok; ++i) {
    Call c;
    c.name = r.str();
    const uint32_t na = r.get<uint32_t>();
    for (uint32_t k = 0; k < na && r.ok; ++k) {
      ArgSpec a;
      a.kind = r.get<uint32_t>(); a.aux = r.get<uint32_t>(); a.value = r.get<uint64_t>();
      c.spec.push_back(a);
    }
    if (c.name == "memset") {
      c.is_memset = true;
      c.fn = nullptr;
      if (na != 2) TFIMM_FAIL(TFIMM_EINVAL, "plan: memset takes (pointer, bytes)");
    } else {
      const auto it = table().find(c.name);
      if (it == table().end()) TFIMM_FAIL(TFIMM_EUNSUP, "plan: no executor entry for %s", c.name.c_str());
      if (it->second.nargs != na) TFIMM_FAIL(TFIMM_EINVAL, "plan: %s takes %zu arguments, blob has %u", c.name.c_str(), it->second.nargs, na);
      c.fn = it->second.fn;
    }
    pl.calls.push_back(std::move(c));
  }
  pl.input_call = r.get<int32_t>();
  pl.stem_call = r.get<int32_t>(); pl.stem_struct = r.get<int32_t>();
  pl.stem_pad_t = r.get<int32_t>(); pl.stem_pad_l = r.get<int32_t>();
  for (uint32_t i = 0, n = r.get<uint32_t>(); i < n && r.ok; ++i) {
    Output o;
    o.name = r.str();
    o.slab = r.get<uint32_t>(); o.offset = r.get<uint64_t>(); o.rows_per_image = r.get<uint64_t>(); o.cols = r.get<uint64_t>();
    o.dtype = r.get<uint32_t>();
    pl.outputs.push_back(o);
  }
  if (!r.ok) TFIMM_FAIL(TFIMM_EINVAL, "plan: truncated blob");
  if (pl.input_call < 0 || pl.input_call >= (int)pl.calls.size()) TFIMM_FAIL(TFIMM_EINVAL, "plan: no input call");
  // everything plan_forward / plan_output index with: checked here, once
  if (pl.calls[pl.input_call].spec.size() < 2) TFIMM_FAIL(TFIMM_EINVAL, "plan: the input call takes (pointer, dtype, ...)");
  if (pl.stem_call >= 0) {
    if (pl.stem_call >= (int)pl.calls.size() || pl.stem_struct < 0 || pl.stem_struct >= (int)pl.structs.size() ||
        pl.structs[pl.stem_struct].bytes.size() < sizeof(tfimm_stem_desc))
      TFIMM_FAIL(TFIMM_EINVAL, "plan: bad stem reference");
  }
  // an output is rows_per_image x batch rows of `cols` elements (bf16 or fp32) starting at `offset`: the whole extent inside its slab
  for (const auto& o : pl.outputs) {
    if (o.slab >= pl.slab_off.size()) TFIMM_FAIL(TFIMM_EINVAL, "plan: output '%s' outside its slab", o.name.c_str());
    const uint64_t cap = pl.slab_bytes[o.slab], esz = o.dtype == 1 ? 4 : 2;
    // every product is checked by DIVISION before it is formed: rows_per_image < 2^40 times an unbounded 32-bit batch, or
    // rows times cols, wrap in 64 bits (rows = cols = 2^32 gives 0 and passed the multiplied form of this check)
    bool ok = o.offset < cap && o.cols > 0 && o.rows_per_image > 0 && pl.batch > 0 && o.cols <= kMaxBuf && o.rows_per_image <= kMaxBuf / pl.batch;
    if (ok) {
      const uint64_t rows = o.rows_per_image * (uint64_t)pl.batch;            // <= kMaxBuf by the division above
      ok = o.cols <= kMaxBuf / rows && rows * o.cols <= (cap - o.offset) / esz;
    }
    if (!ok) TFIMM_FAIL(TFIMM_EINVAL, "plan: output '%s' outside its slab", o.name.c_str());
  }
  // a descriptor argument must be at least as large as the struct its entry point reads
  for (const auto& c : pl.calls) {
    if (c.is_memset) continue;
    const auto& e = table().find(c.name)->second;
    for (size_t k = 0; k < c.spec.size(); ++k)
      if (c.spec[k].kind == A_STRUCT) {
        if (c.spec[k].aux >= pl.structs.size()) TFIMM_FAIL(TFIMM_EINVAL, "plan: bad struct reference");
        if (e.desc_bytes && pl.structs[c.spec[k].aux].bytes.size() < e.desc_bytes)
          TFIMM_FAIL(TFIMM_EINVAL, "plan: %s descriptor has %zu bytes, the entry point reads %zu", c.name.c_str(),
                     pl.structs[c.spec[k].aux].bytes.size(), e.desc_bytes);
      }
  }
  return 0;
}

int resolve(const Plan& pl, uint32_t kind, uint32_t aux, uint64_t value, void** out) {
  switch (kind) {
    case A_NULL: *out = nullptr; return 0;
    case A_INPUT: *out = nullptr; return 0;                      // patched per forward
    case A_SLAB:
      if (aux >= pl.slab_off.size() || value > pl.slab_bytes[aux]) TFIMM_FAIL(TFIMM_EINVAL, "plan: bad slab reference");
      *out = pl.ws + pl.slab_off[aux] + value;
      return 0;
    case A_CONST:
      if (aux >= pl.const_off.size() || value > pl.const_bytes[aux]) TFIMM_FAIL(TFIMM_EINVAL, "plan: bad constant reference");
      *out = pl.ws + pl.const_off[aux] + value;
      return 0;
    case A_HOST:
      if (aux >= pl.host_consts.size() || value > pl.host_consts[aux].size()) TFIMM_FAIL(TFIMM_EINVAL, "plan: bad host constant reference");
      *out = const_cast<uint8_t*>(pl.host_consts[aux].data()) + value;
      return 0;
    default: TFIMM_FAIL(TFIMM_EINVAL, "plan: pointer kind %u", kind);
  }
}

}  // namespace

extern "C" {

int tfimm_hip_plan_query(const void* blob, size_t bytes, tfimm_plan_info* info) {
  if (!blob || !info) TFIMM_FAIL(TFIMM_EINVAL, "plan_query: null pointer");
  Plan pl;
  const int rc = parse(blob, bytes, pl);
  if (rc != 0) return rc;
  memset(info, 0, sizeof(*info));
  info->workspace_bytes = pl.workspace_bytes;
  info->batch = (int32_t)pl.batch;
  info->in_h = pl.img_h; info->in_w = pl.img_w; info->in_c = pl.img_c;
  info->n_calls = (int32_t)pl.calls.size();
  info->n_outputs = (int32_t)pl.outputs.size();
  return 0;
}

int tfimm_hip_plan_create(const void* blob, size_t bytes, void* workspace, void* stream, tfimm_plan_t* out) {
  if (!blob || !workspace || !out) TFIMM_FAIL(TFIMM_EINVAL, "plan_create: null pointer");
  if ((uintptr_t)workspace & 255) TFIMM_FAIL(TFIMM_EINVAL, "plan_create: the workspace must be 256-byte aligned");
  Plan* pl = new Plan();
  int rc = parse(blob, bytes, *pl);
  if (rc != 0) { delete pl; return rc; }
  pl->ws = static_cast<char*>(workspace);
  for (size_t i = 0; i < pl->const_off.size(); ++i) {
    const hipError_t e = hipMemcpyAsync(pl->ws + pl->const_off[i], static_cast<const uint8_t*>(blob) + pl->const_src[i], pl->const_bytes[i],
                                        hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { delete pl; tfimm_set_error("plan_create: constant upload failed: %s", hipGetErrorString(e)); return (int)e; }
  }
  // the blob may be released after this call: wait for the uploads
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) { delete pl; tfimm_set_error("plan_create: %s", hipGetErrorString(e)); return (int)e; }
  for (auto& s : pl->structs)
    for (const auto& rl : s.relocs) {
      void* p = nullptr;
      rc = resolve(*pl, rl.kind, rl.aux, rl.value, &p);
      if (rc != 0) { delete pl; return rc; }
      memcpy(s.bytes.data() + rl.field_offset, &p, sizeof(p));
    }
  for (auto& c : pl->calls) {
    c.args.resize(c.spec.size());
    for (size_t k = 0; k < c.spec.size(); ++k) {
      const ArgSpec& a = c.spec[k];
      Arg v{0, 0.0, nullptr};
      if (a.kind == A_INT) { v.i = (int64_t)a.value; v.f = (double)v.i; }
      else if (a.kind == A_FLOAT) { memcpy(&v.f, &a.value, 8); v.i = (int64_t)v.f; }
      else if (a.kind == A_STRUCT) {
        if (a.aux >= pl->structs.size()) { delete pl; TFIMM_FAIL(TFIMM_EINVAL, "plan: bad struct reference"); }
        v.p = pl->structs[a.aux].bytes.data();
      } else {
        rc = resolve(*pl, a.kind, a.aux, a.value, &v.p);
        if (rc != 0) { delete pl; return rc; }
      }
      c.args[k] = v;
    }
  }
  *out = pl;
  return 0;
}

int tfimm_hip_plan_forward(tfimm_plan_t plan, const void* input, int in_dtype, void* stream) {
  Plan* pl = static_cast<Plan*>(plan);
  if (!pl || !input) TFIMM_FAIL(TFIMM_EINVAL, "plan_forward: null pointer");
  if (in_dtype != 0 && in_dtype != 1) TFIMM_FAIL(TFIMM_EINVAL, "plan_forward: in_dtype 0 = float32, 1 = bf16");
  bool skip_cast = false;
  if (pl->stem_call >= 0) {
    // the fused ResNet stem reads the caller's RGB image itself (border, 4th channel and rounding applied while it fills its
    // LDS ring): no conversion pass
    tfimm_stem_desc* d = reinterpret_cast<tfimm_stem_desc*>(pl->structs[pl->stem_struct].bytes.data());
    d->x = input;
    d->in_dtype = in_dtype == 1 ? 1 : 2;
    d->H = pl->img_h; d->W = pl->img_w; d->pad_t = pl->stem_pad_t; d->pad_l = pl->stem_pad_l;
    skip_cast = true;
  }
  for (size_t i = 0; i < pl->calls.size(); ++i) {
    Call& c = pl->calls[i];
    if ((int)i == pl->input_call) {
      if (skip_cast) continue;
      c.args[0].p = const_cast<void*>(input);
      c.args[1].i = in_dtype;
    }
    int rc;
    if (c.is_memset) rc = tfimm_hip_memset_async(c.args[0].p, 0, (size_t)c.args[1].i, stream);
    else rc = c.fn(c.args.data(), stream);
    if (rc != 0) return rc;
  }
  return 0;
}

int tfimm_hip_plan_output(tfimm_plan_t plan, const char* name, void** ptr, int64_t* rows, int64_t* cols, int* dtype) {
  Plan* pl = static_cast<Plan*>(plan);
  if (!pl || !name || !ptr) TFIMM_FAIL(TFIMM_EINVAL, "plan_output: null pointer");
  for (const auto& o : pl->outputs)
    if (o.name == name) {
      *ptr = pl->ws + pl->slab_off[o.slab] + o.offset;
      if (rows) *rows = (int64_t)o.rows_per_image * pl->batch;
      if (cols) *cols = (int64_t)o.cols;
      if (dtype) *dtype = (int)o.dtype;
      return 0;
    }
  TFIMM_FAIL(TFIMM_EINVAL, "plan_output: the plan has no output named '%s'", name);
}

int tfimm_hip_plan_destroy(tfimm_plan_t plan) {
  delete static_cast<Plan*>(plan);
  return 0;
}

}  // extern "C"
