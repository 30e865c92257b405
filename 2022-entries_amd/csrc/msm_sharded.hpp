// msm_sharded.hpp -- one MSM over several MI355X from ONE process, behind the same C ABI (included by msm_engine.hip).
//
// The reference is single-device everywhere (SPK msm/pippenger.cuh:400-416 hard-codes device 0), but its harness talks to the
// library through mult_pippenger_init / mult_pippenger_inf or MSMPreprocessPoints / MSMRun (P1A 6block/src/lib.rs:54-109,
// CMB MSM.h:72-75) -- so the multi-GPU path has to live BEHIND those calls for a harness to use it unchanged.  A sharded
// context owns one ordinary context per device; the host-side orchestration is the multi-stream pattern of
// P1A matter-labs/src/lib.rs:125-201 with devices in the place of streams:
//
//   set_bases   shard g receives the contiguous slice [g*ceil(n/G), ...) of the bases, uploaded concurrently (one host thread
//               per device);
//   run         every shard runs the full single-device pipeline on its slice of every scalar batch and yields one partial
//               point per batch, which its host thread has ALREADY folded out of W window sums on the host -- so in this
//               one-process form the G partials are in host memory when the shards return and "the final 8-point curve add"
//               is a host fold of G x batches x 144 B.  No collective is needed for that, and none is run by default.
//               Option "combine" = 2 additionally routes the partials through ONE ncclAllGather over RCCL/xGMI
//               (single-process communicator, ncclCommInitAll at set_bases) and requires the exchanged copy to equal what
//               was sent: a link check for bring-up on a new node, not a step the result depends on.  The collective that IS
//               needed -- partials living in different processes -- is dist.py's all_gather (one process per GPU).
//
// `devices` may name a device more than once (logical shards on one GPU): that is how the 2^28 / 8-shard workload is rehearsed
// on a one-GPU box.
#pragma once

// --- RCCL, loaded at run time (the library must stay loadable where librccl is absent; torch ships its own copy) ----------
struct RcclState {
  typedef int (*comm_init_all_t)(void** comms, int ndev, const int* devlist);
  typedef int (*comm_destroy_t)(void* comm);
  typedef int (*all_gather_t)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st);
  typedef int (*group_t)();
  typedef const char* (*err_t)(int);
  void* lib = nullptr;
  comm_init_all_t comm_init_all = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  all_gather_t all_gather = nullptr;
  group_t group_start = nullptr, group_end = nullptr;
  err_t error_string = nullptr;
  std::vector<void*> comms;          // one per shard
  std::vector<DevBuf> send, recv;    // per shard, on its device
  std::vector<hipStream_t> streams;
  void* host_recv = nullptr;         // pinned
  size_t host_recv_bytes = 0;
  std::string why_not;               // reason the RCCL path is unavailable, for error messages
};

static const int kNcclUint8 = 1;     // ncclDataType_t::ncclUint8 (rccl.h)

static bool rccl_load(RcclState& r) {
  const char* env = getenv("MI355_MSM_RCCL_LIB");
  const char* names[] = {env, "librccl.so", "librccl.so.1"};
  // prefer a copy that is already mapped (torch brings its own librccl.so): two RCCL instances in one process are two worlds
  for (const char* n : names)
    if (n && *n && (r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  for (const char* n : names)
    if (!r.lib && n && *n) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) {
    r.why_not = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?");
    return false;
  }
  r.comm_init_all = (RcclState::comm_init_all_t)dlsym(r.lib, "ncclCommInitAll");
  r.comm_destroy = (RcclState::comm_destroy_t)dlsym(r.lib, "ncclCommDestroy");
  r.all_gather = (RcclState::all_gather_t)dlsym(r.lib, "ncclAllGather");
  r.group_start = (RcclState::group_t)dlsym(r.lib, "ncclGroupStart");
  r.group_end = (RcclState::group_t)dlsym(r.lib, "ncclGroupEnd");
  r.error_string = (RcclState::err_t)dlsym(r.lib, "ncclGetErrorString");
  if (!r.comm_init_all || !r.comm_destroy || !r.all_gather || !r.group_start || !r.group_end) {
    r.why_not = "librccl lacks ncclCommInitAll / ncclAllGather / ncclGroupStart";
    return false;
  }
  return true;
}

#define RCCL_OK(r, expr)                                                                             \
  do {                                                                                               \
    int rc_ = (expr);                                                                                \
    if (rc_ != 0) {                                                                                  \
      char buf_[384];                                                                                \
      snprintf(buf_, sizeof buf_, "%s failed: %s", #expr, (r).error_string ? (r).error_string(rc_) : "rccl error"); \
      throw HipFailure(-1, buf_);                                                                    \
    }                                                                                                \
  } while (0)

// "0,1,2" | "all" | "0-7"
static std::vector<int> parse_device_list(const char* text) {
  std::vector<int> devs;
  std::string t(text);
  if (t == "all" || t == "ALL") {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw std::runtime_error("MI355_MSM_DEVICES=all: no HIP device visible");
    for (int i = 0; i < count; i++) devs.push_back(i);
    return devs;
  }
  size_t pos = 0;
  while (pos <= t.size()) {
    size_t comma = t.find(',', pos);
    if (comma == std::string::npos) comma = t.size();
    std::string item = t.substr(pos, comma - pos);
    if (item.empty()) throw std::runtime_error("MI355_MSM_DEVICES: empty entry in '" + t + "'");
    const size_t dash = item.find('-');
    char* endp = nullptr;
    if (dash != std::string::npos && dash > 0) {
      if (dash + 1 >= item.size()) throw std::runtime_error("MI355_MSM_DEVICES: cannot parse '" + item + "'");
      const long a = strtol(item.substr(0, dash).c_str(), &endp, 10);
      if (*endp) throw std::runtime_error("MI355_MSM_DEVICES: cannot parse '" + item + "'");
      const long b = strtol(item.substr(dash + 1).c_str(), &endp, 10);
      if (*endp || b < a || a < 0 || b > 1023) throw std::runtime_error("MI355_MSM_DEVICES: cannot parse '" + item + "'");
      for (long d = a; d <= b; d++) devs.push_back((int)d);
    } else {
      const long a = strtol(item.c_str(), &endp, 10);
      if (*endp || a < 0 || a > 1023) throw std::runtime_error("MI355_MSM_DEVICES: cannot parse '" + item + "'");
      devs.push_back((int)a);
    }
    pos = comma + 1;
  }
  if (devs.empty() || devs.size() > 64) throw std::runtime_error("MI355_MSM_DEVICES: need 1..64 devices");
  return devs;
}

// Contiguous slice of range(n) owned by shard g of G (dist.py::shard_bounds is the Python twin).
static inline void shard_bounds(size_t n, size_t G, size_t g, size_t& lo, size_t& hi) {
  const size_t per = (n + G - 1) / G;
  lo = std::min(n, g * per);
  hi = std::min(n, lo + per);
}

static void take(RustError e) {
  if (!e.code) return;
  std::string m = e.message ? e.message : "";
  free(e.message);
  throw HipFailure(e.code, m);
}

// Run fn(g) for every shard on its own host thread; rethrow the first failure with the shard named (host_pipeline.hpp).
template <class Fn>
static void for_each_shard(mi355_msm_ctx* ctx, Fn&& fn) {
  msm_host::run_on_shards(ctx->shards.size(), fn, [&](size_t g) {
    char buf[64];
    snprintf(buf, sizeof buf, "shard %zu (device %d)", g, ctx->shards[g]->device);
    return std::string(buf);
  });
}

static void sharded_destroy(mi355_msm_ctx* ctx) {
  if (ctx->rccl) {
    RcclState& r = *ctx->rccl;
    for (size_t g = 0; g < r.comms.size(); g++) {
      (void)hipSetDevice(ctx->shards[g]->device);
      if (r.comms[g]) (void)r.comm_destroy(r.comms[g]);
      if (g < r.send.size()) r.send[g].release();
      if (g < r.recv.size()) r.recv[g].release();
      if (g < r.streams.size() && r.streams[g]) (void)hipStreamDestroy(r.streams[g]);
    }
    if (r.host_recv) (void)hipHostFree(r.host_recv);
    delete ctx->rccl;
    ctx->rccl = nullptr;
  }
  for (mi355_msm_ctx* sh : ctx->shards) {
    RustError d = mi355_msm_destroy(sh);
    if (d.message) free(d.message);
  }
  ctx->shards.clear();
}

static mi355_msm_ctx* sharded_create(int curve, const int* devices, int ndevices) {
  mi355_msm_ctx* ctx = new mi355_msm_ctx();
  ctx->curve = curve;
  ctx->device = devices[0];
  try {
    for (int g = 0; g < ndevices; g++) {
      mi355_msm_ctx* sh = nullptr;
      take(mi355_msm_create(&sh, curve, devices[g]));
      ctx->shards.push_back(sh);
    }
  } catch (...) {
    sharded_destroy(ctx);
    delete ctx;
    throw;
  }
  ctx->shard_lo.assign((size_t)ndevices + 1, 0);
  return ctx;
}

static int pointer_device(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return attr.type == hipMemoryTypeDevice ? attr.device : -1;
}

static bool rccl_ready(mi355_msm_ctx* ctx, size_t payload);

// Bases for the whole sharded context: host memory (or CanonicalSerialize records), or ONE device buffer (any device: a shard
// on another GPU pulls its slice over xGMI into a staging buffer first).
static void sharded_set_bases(mi355_msm_ctx* ctx, const void* data, size_t n, size_t stride, bool serialized, bool on_device) {
  const size_t G = ctx->shards.size();
  for (size_t g = 0; g <= G; g++) {
    size_t lo, hi;
    shard_bounds(n, G, std::min(g, G - 1), lo, hi);
    ctx->shard_lo[g] = g < G ? lo : n;
  }
  int src_dev = -1;
  if (on_device) {
    src_dev = pointer_device(data);
    if (src_dev >= 0) {
      HIP_OK(hipSetDevice(src_dev));
      HIP_OK(hipDeviceSynchronize());   // the producer may have written the buffer on another stream
    }
  }
  ctx->nbases = 0;
  for_each_shard(ctx, [&](size_t g) {
    mi355_msm_ctx* sh = ctx->shards[g];
    const size_t lo = ctx->shard_lo[g], cnt = ctx->shard_lo[g + 1] - lo;
    const uint8_t* src = (const uint8_t*)data + lo * stride;
    if (!on_device) {
      set_bases_host(sh, cnt ? src : nullptr, cnt, stride, serialized);
      return;
    }
    ensure_device(sh);
    // ("force_peer_staging": test hook -- logical shards share one device, so the branch below would otherwise first execute on the
    //  first multi-GPU box; the copy is then device-local, everything else is the code a peer pull runs)
    if ((src_dev == sh->device && !ctx->opt_force_peer_staging) || cnt == 0) {
      set_bases_device(sh, src, cnt, stride);
      return;
    }
    __atomic_fetch_add(&ctx->peer_stagings, 1, __ATOMIC_RELAXED);
    DevBuf stage;
    try {
      stage.reserve(cnt * stride);
      // ON THE SHARD'S STREAM: a device-to-device hipMemcpy is not synchronous with the host and lives on the null stream, which the
      // shard's non-blocking stream does not wait for -- the conversion kernel read the staging buffer before the copy had landed
      // (found by tests/test_gpu_sharded.py::test_peer_staging_branches_on_logical_shards in round 5, on the first box where the
      // timing exposed it: this branch had never executed before the "force_peer_staging" hook existed)
      HIP_OK(hipMemcpyAsync(stage.p, src, cnt * stride, hipMemcpyDefault, sh->own_stream));
      set_bases_device(sh, stage.p, cnt, stride);
    } catch (...) {
      stage.release();
      throw;
    }
    stage.release();
  });
  ctx->nbases = n;
  if (ctx->opt_combine == 2) (void)rccl_ready(ctx, 3 * coord_bytes(ctx->curve));   // communicator + buffers now, not inside the first run
}

static bool devices_distinct(const mi355_msm_ctx* ctx) {
  for (size_t a = 0; a < ctx->shards.size(); a++)
    for (size_t b = a + 1; b < ctx->shards.size(); b++)
      if (ctx->shards[a]->device == ctx->shards[b]->device) return false;
  return true;
}

// combine = 2 only: load librccl, ncclCommInitAll over the shard devices (first called from set_bases, so that the
// communicator is not built inside a timed run), per-shard send/recv buffers.  false = host fold only.
static bool rccl_ready(mi355_msm_ctx* ctx, size_t payload) {
  if (ctx->opt_combine != 2) return false;
  if (!ctx->rccl) {
    ctx->rccl = new RcclState();
    RcclState& r = *ctx->rccl;
    if (!devices_distinct(ctx)) {
      r.why_not = "a device is listed more than once (logical shards): an RCCL communicator needs distinct devices";
    } else if (rccl_load(r)) {
      const size_t G = ctx->shards.size();
      std::vector<int> devs(G);
      for (size_t g = 0; g < G; g++) devs[g] = ctx->shards[g]->device;
      r.comms.assign(G, nullptr);
      const int rc = r.comm_init_all(r.comms.data(), (int)G, devs.data());
      if (rc != 0) {
        r.why_not = std::string("ncclCommInitAll failed: ") + (r.error_string ? r.error_string(rc) : "?");
        r.comms.clear();
      } else {
        r.send.resize(G);
        r.recv.resize(G);
        r.streams.assign(G, nullptr);
        for (size_t g = 0; g < G; g++) {
          HIP_OK(hipSetDevice(devs[g]));
          HIP_OK(hipStreamCreateWithFlags(&r.streams[g], hipStreamNonBlocking));
        }
      }
    }
  }
  RcclState& r = *ctx->rccl;
  if (r.comms.empty()) {
    if (ctx->opt_combine == 2) throw HipFailure(-1, "combine = 2 (require RCCL): " + r.why_not);
    return false;
  }
  const size_t G = ctx->shards.size();
  for (size_t g = 0; g < G; g++) {
    HIP_OK(hipSetDevice(ctx->shards[g]->device));
    r.send[g].reserve(payload);
    r.recv[g].reserve(G * payload);
  }
  if (r.host_recv_bytes < G * payload) {
    if (r.host_recv) (void)hipHostFree(r.host_recv);
    HIP_OK(hipHostMalloc(&r.host_recv, G * payload, hipHostMallocDefault));
    r.host_recv_bytes = G * payload;
  }
  return true;
}

// partials: [G][batches][pb] on the host.  Exchanges them with ONE all-gather per run and returns the gathered copy as seen
// by shard 0 (G x payload bytes, pinned host memory).
static const uint8_t* rccl_exchange(mi355_msm_ctx* ctx, const uint8_t* partials, size_t payload) {
  RcclState& r = *ctx->rccl;
  const size_t G = ctx->shards.size();
  for (size_t g = 0; g < G; g++) {
    HIP_OK(hipSetDevice(ctx->shards[g]->device));
    HIP_OK(hipMemcpyAsync(r.send[g].p, partials + g * payload, payload, hipMemcpyHostToDevice, r.streams[g]));
  }
  RCCL_OK(r, r.group_start());
  for (size_t g = 0; g < G; g++)
    RCCL_OK(r, r.all_gather(r.send[g].p, r.recv[g].p, payload, kNcclUint8, r.comms[g], r.streams[g]));
  RCCL_OK(r, r.group_end());
  HIP_OK(hipSetDevice(ctx->shards[0]->device));
  HIP_OK(hipMemcpyAsync(r.host_recv, r.recv[0].p, G * payload, hipMemcpyDeviceToHost, r.streams[0]));
  for (size_t g = 0; g < G; g++) {
    HIP_OK(hipSetDevice(ctx->shards[g]->device));
    HIP_OK(hipStreamSynchronize(r.streams[g]));
  }
  ctx->rccl_exchanges++;
  return (const uint8_t*)r.host_recv;
}

static void sharded_run(mi355_msm_ctx* ctx, void* out, const void* scalars, size_t n, size_t batches, bool on_device, hipStream_t stream) {
  if (n > ctx->nbases) bad_arg("npoints %zu exceeds the %zu uploaded bases", n, ctx->nbases);
  if (!out) bad_arg("null output pointer");
  const size_t G = ctx->shards.size();
  const size_t pb = 3 * coord_bytes(ctx->curve);   // one Projective image
  const size_t payload = batches * pb;
  int src_dev = -1;
  if (on_device && n * batches) {
    src_dev = pointer_device(scalars);
    // the scalars become ready on the caller's stream; the shards work on their own streams (and devices)
    if (src_dev >= 0) HIP_OK(hipSetDevice(src_dev));
    HIP_OK(hipStreamSynchronize(stream));
  }
  std::vector<uint8_t> partials(G * payload + 1);
  for_each_shard(ctx, [&](size_t g) {
    mi355_msm_ctx* sh = ctx->shards[g];
    const size_t lo = ctx->shard_lo[g], cap = ctx->shard_lo[g + 1] - lo;
    const size_t cnt = n > lo ? std::min(cap, n - lo) : 0;   // a run may use a prefix of the bases
    uint8_t* po = partials.data() + g * payload;
    const uint8_t* src = (const uint8_t*)scalars + lo * 32;
    if (!on_device) {
      run_host(sh, po, src, cnt, batches, n);
      return;
    }
    ensure_device(sh);
    if ((src_dev == sh->device && !ctx->opt_force_peer_staging) || cnt == 0) {
      run_device(sh, po, src, cnt, batches, n, sh->own_stream);
      return;
    }
    __atomic_fetch_add(&ctx->peer_stagings, 1, __ATOMIC_RELAXED);
    // another GPU's memory: pull this shard's slice of every batch over xGMI, then run on the local copy
    sh->scalars.reserve(std::max<size_t>(cnt * batches * 32, 32));
    for (size_t b = 0; b < batches; b++)
      HIP_OK(hipMemcpyAsync(sh->scalars.as<uint8_t>() + b * cnt * 32, src + b * n * 32, cnt * 32, hipMemcpyDefault, sh->own_stream));
    run_device(sh, po, sh->scalars.p, cnt, batches, cnt, sh->own_stream);
  });
  // exchange + fold
  const uint8_t* gathered = nullptr;
  if (G > 0 && payload && rccl_ready(ctx, payload)) {
    gathered = rccl_exchange(ctx, partials.data(), payload);
    // the host already holds what was sent: the exchanged copy must be those very bytes
    if (memcmp(gathered, partials.data(), G * payload) != 0) throw HipFailure(-1, "RCCL all-gather returned partials that differ from the ones sent");
  }
  const uint8_t* src = gathered ? gathered : partials.data();
  std::vector<uint8_t> col(G * pb);
  for (size_t b = 0; b < batches; b++) {
    for (size_t g = 0; g < G; g++) memcpy(col.data() + g * pb, src + g * payload + b * pb, pb);
    take(mi355_msm_fold(ctx->curve, (uint8_t*)out + b * pb, col.data(), G));
  }
}
