// Multi-GPU (one process per GPU) emulated GEMM behind the C ABI of include/gemmul8_dist.h.
//
// No counterpart in the reference (no multi-GPU code); this is the sharded form of its pipeline drivers
// (GEMMul8/src/gemmul8_real.hpp:52-211, gemmul8_complex.hpp:52-226) that BASELINE.json's north_star asks for: host C++ that
// partitions the path, calls the single-GPU phase entry points of gemmul8_c.h on each rank's share and moves the small
// coupling data with RCCL over xGMI.  Host code only: no kernels here, no BLAS, no torch.
//
//   blocks   : rank (i, j) of a Gr x Gc grid runs every modulus on C[rows_i, cols_j]; accurate mode adds ONE
//              all-reduce(MAX) of int32[m + n] (row maxima need all column blocks, column maxima all row blocks).
//   moduli   : rank r multiplies moduli [t0_r, t1_r); INT8 residue blocks travel point-to-point (one grouped
//              ncclSend/ncclRecv set, every xGMI link busy at once -- xGMI is point-to-point, a ring would use 2 of 7 links);
//              CRT in reference order on the rank's columns.
//   fp64sum  : the same with the FP64 partial-sum reduce-scatter north_star names instead of the residue exchange.
#include "../../include/gemmul8_dist.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <netdb.h>
#include <netinet/in.h>
#include <rccl/rccl.h>  // types and enums only: librccl is opened at run time
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------ RCCL, bound at run time
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // optional: read-back of the communicator size
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

// The RCCL that belongs to the HIP runtime this process already uses: the copy that is mapped (PyTorch ships its own next to
// its libamdhip64 -- a second HIP runtime pulled in by /opt/rocm's librccl would not share devices or pointers with the first),
// otherwise the system library.
void* open_rccl() {
    if (FILE* f = std::fopen("/proc/self/maps", "r")) {
        char line[1024];
        void* h = nullptr;
        while (!h && std::fgets(line, sizeof line, f)) {
            if (!std::strstr(line, "librccl.so")) continue;
            char* path = std::strchr(line, '/');
            if (!path) continue;
            path[std::strcspn(path, "\n")] = 0;
            h = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        }
        std::fclose(f);
        if (h) return h;
    }
    for (const char* name : {"librccl.so.1", "librccl.so"})
        if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) return h;
    return nullptr;
}

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        void* h = open_rccl();
        if (!h) {
            std::fprintf(stderr, "[GEMMUL8 DIST] librccl not found: %s\n", dlerror());
            return x;
        }
#define OZ2_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(h, name))
        OZ2_SYM(GetUniqueId, "ncclGetUniqueId");
        OZ2_SYM(CommInitRank, "ncclCommInitRank");
        OZ2_SYM(CommDestroy, "ncclCommDestroy");
        OZ2_SYM(CommCount, "ncclCommCount");
        OZ2_SYM(AllReduce, "ncclAllReduce");
        OZ2_SYM(ReduceScatter, "ncclReduceScatter");
        OZ2_SYM(Send, "ncclSend");
        OZ2_SYM(Recv, "ncclRecv");
        OZ2_SYM(GroupStart, "ncclGroupStart");
        OZ2_SYM(GroupEnd, "ncclGroupEnd");
        OZ2_SYM(GetErrorString, "ncclGetErrorString");
#undef OZ2_SYM
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.ReduceScatter && x.Send && x.Recv && x.GroupStart &&
               x.GroupEnd;
        if (!x.ok) std::fprintf(stderr, "[GEMMUL8 DIST] librccl lacks a required entry point\n");
        return x;
    }();
    return r;
}

int nccl_status(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    std::fprintf(stderr, "[GEMMUL8 DIST] %s failed: %s\n", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    return 1000 + (int)r;  // positive, like a runtime error of the C ABI
}

struct RcclCtx {
    ncclComm_t comm = nullptr;
};

int rccl_allreduce_max_i32(void* ctx, void* buf, size_t count, void* stream) {
    return nccl_status(rccl().AllReduce(buf, buf, count, ncclInt32, ncclMax, static_cast<RcclCtx*>(ctx)->comm, (hipStream_t)stream), "ncclAllReduce");
}
int rccl_sendrecv(void* ctx, int nops, const gemmul8_p2p_op* ops, void* stream) {
    if (nops <= 0) return 0;
    const Rccl& R = rccl();
    ncclComm_t comm = static_cast<RcclCtx*>(ctx)->comm;
    int rc = nccl_status(R.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < nops && !rc; ++i) {
        if (ops[i].bytes == 0) continue;
        rc = ops[i].is_send ? nccl_status(R.Send(ops[i].buf, ops[i].bytes, ncclInt8, ops[i].peer, comm, (hipStream_t)stream), "ncclSend")
                            : nccl_status(R.Recv(ops[i].buf, ops[i].bytes, ncclInt8, ops[i].peer, comm, (hipStream_t)stream), "ncclRecv");
    }
    const int rc2 = nccl_status(R.GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc2;
}
int rccl_reduce_scatter_sum_f64(void* ctx, const void* send, void* recv, size_t recv_count, void* stream) {
    return nccl_status(rccl().ReduceScatter(send, recv, recv_count, ncclDouble, ncclSum, static_cast<RcclCtx*>(ctx)->comm, (hipStream_t)stream),
                       "ncclReduceScatter");
}
void rccl_destroy(void* ctx) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    if (c && c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

// ------------------------------------------------------------------------------------------------ TCP hand-off of the unique id
bool send_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w <= 0) return false;
        c += w, n -= (size_t)w;
    }
    return true;
}
bool recv_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t r = ::recv(fd, c, n, 0);
        if (r <= 0) return false;
        c += r, n -= (size_t)r;
    }
    return true;
}

// Rank 0's listening address.  MASTER_ADDR names rank 0 for the OTHER ranks; on rank 0 itself it may resolve to an address other nodes
// cannot reach: Debian / Ubuntu map the host's own name to 127.0.1.1 in /etc/hosts, and every 127/8 address is locally bindable, so a
// bind to the resolved address "succeeds" and listens on loopback only (a multi-node rendezvous then times out).
// Policy: a non-loopback address or the canonical 127.0.0.1 (a deliberately single-node job) is bound as given, falling back to all
// interfaces when the host does not own it (service / NAT address); any OTHER 127/8 result is the hostname alias -> all interfaces
// first, as torch's TCPStore does.  Strangers that reach the port are turned away by the challenge below.
//
// Hand-off protocol (version 2): on accept rank 0 sends a 16-byte random salt; the client answers with a 24-byte hello
// {magic, rank, world, reserved, mac} where mac = SipHash-2-4(key; salt | rank | world | port) and key is 128 bits derived from
// GEMMUL8_DIST_SECRET (a job secret the launcher may export to every rank).  The secret itself never travels and a recorded hello cannot be
// replayed (fresh salt per connection).  WITHOUT GEMMUL8_DIST_SECRET the key derives from MASTER_PORT alone: then this is only a filter
// against stray connections, not authentication -- anyone who knows the port can compute it.  A rank is served once; a rank that asks AGAIN
// (its first reply may have been lost to its 5 s receive timeout) is served one more time, no more.  Every accept / recv has a deadline: a
// missing rank ends in an error message after GEMMUL8_DIST_TIMEOUT seconds (default 120) instead of a silent hang inside ncclCommInitRank.
struct IdHello {
    uint32_t magic, rank, world, reserved;
    uint64_t mac;
};
static_assert(sizeof(IdHello) == 24, "hello is 24 bytes on the wire");
constexpr uint32_t kHelloMagic = 0x324c5547u;  // "GUL2"
struct IdKey {
    uint64_t k0, k1;
};
inline uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
// SipHash-2-4 (Aumasson & Bernstein), 64-bit tag over `n` bytes
uint64_t siphash24(const IdKey& key, const unsigned char* in, size_t n) {
    uint64_t v0 = 0x736f6d6570736575ull ^ key.k0, v1 = 0x646f72616e646f6dull ^ key.k1, v2 = 0x6c7967656e657261ull ^ key.k0, v3 = 0x7465646279746573ull ^ key.k1;
    auto round = [&] {
        v0 += v1, v1 = rotl64(v1, 13), v1 ^= v0, v0 = rotl64(v0, 32);
        v2 += v3, v3 = rotl64(v3, 16), v3 ^= v2;
        v0 += v3, v3 = rotl64(v3, 21), v3 ^= v0;
        v2 += v1, v1 = rotl64(v1, 17), v1 ^= v2, v2 = rotl64(v2, 32);
    };
    const size_t full = n / 8 * 8;
    for (size_t i = 0; i < full; i += 8) {
        uint64_t mword;
        std::memcpy(&mword, in + i, 8);
        v3 ^= mword, round(), round(), v0 ^= mword;
    }
    uint64_t last = (uint64_t)(n & 0xff) << 56;
    for (size_t i = full; i < n; ++i) last |= (uint64_t)in[i] << (8 * (i - full));
    v3 ^= last, round(), round(), v0 ^= last;
    v2 ^= 0xff;
    round(), round(), round(), round();
    return v0 ^ v1 ^ v2 ^ v3;
}
IdKey id_key_from_env(int master_port) {
    IdKey key{0x9e3779b97f4a7c15ull ^ (uint64_t)master_port, 0xc2b2ae3d27d4eb4full + (uint64_t)master_port};
    if (const char* sec = std::getenv("GEMMUL8_DIST_SECRET")) {
        uint64_t h0 = 14695981039346656037ull, h1 = 0x84222325cbf29ce4ull;  // two FNV-1a-64 chains with different offsets
        for (const char* c = sec; *c; ++c) {
            h0 = (h0 ^ (unsigned char)*c) * 1099511628211ull;
            h1 = (h1 ^ (unsigned char)*c ^ 0x5c) * 1099511628211ull;
            h1 = rotl64(h1, 29) + h0;
        }
        key.k0 ^= h0, key.k1 ^= h1;
    }
    return key;
}
uint64_t id_mac(const IdKey& key, const unsigned char salt[16], uint32_t rank, uint32_t world, uint32_t port) {
    unsigned char msg[28];
    std::memcpy(msg, salt, 16);
    std::memcpy(msg + 16, &rank, 4);
    std::memcpy(msg + 20, &world, 4);
    std::memcpy(msg + 24, &port, 4);
    return siphash24(key, msg, sizeof msg);
}
void random_salt(unsigned char salt[16]) {
    std::random_device rd;  // /dev/urandom on Linux
    for (int i = 0; i < 16; i += 4) {
        const uint32_t v = rd();
        std::memcpy(salt + i, &v, 4);
    }
}
void set_io_timeout(int fd, int seconds) {
    timeval tv{};
    tv.tv_sec = seconds;
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}
int rendezvous_timeout_s() {
    const char* s = std::getenv("GEMMUL8_DIST_TIMEOUT");
    const int v = s ? std::atoi(s) : 0;
    return v > 0 ? v : 120;
}
// true: bind all interfaces before trying the resolved address (see the policy above)
bool bind_any_first(const sockaddr* resolved) {
    if (resolved->sa_family != AF_INET) return true;
    const uint32_t a = ntohl(reinterpret_cast<const sockaddr_in*>(resolved)->sin_addr.s_addr);
    return (a >> 24) == 127 && a != INADDR_LOOPBACK;
}

int exchange_id_tcp(const char* addr, int port, int rank, int world, const IdKey& key, ncclUniqueId* id) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string ports = std::to_string(port);
    if (getaddrinfo(addr, ports.c_str(), &hints, &res) != 0 || !res) {
        std::fprintf(stderr, "[GEMMUL8 DIST] cannot resolve MASTER_ADDR %s\n", addr);
        return GEMMUL8_E_ARG;
    }
    const int limit = rendezvous_timeout_s();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(limit);
    int rc = GEMMUL8_E_INTERNAL;
    if (rank == 0) {
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        int one = 1;
        if (ls >= 0) setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in any{};
        any.sin_family = AF_INET;
        any.sin_addr.s_addr = htonl(INADDR_ANY);
        any.sin_port = htons((uint16_t)port);
        const sockaddr* first = bind_any_first(res->ai_addr) ? reinterpret_cast<const sockaddr*>(&any) : res->ai_addr;
        const sockaddr* second = first == res->ai_addr ? reinterpret_cast<const sockaddr*>(&any) : res->ai_addr;
        const socklen_t len1 = first == res->ai_addr ? (socklen_t)res->ai_addrlen : (socklen_t)sizeof any;
        const socklen_t len2 = second == res->ai_addr ? (socklen_t)res->ai_addrlen : (socklen_t)sizeof any;
        const bool bound = ls >= 0 && (::bind(ls, first, len1) == 0 || ::bind(ls, second, len2) == 0);
        if (!bound || ::listen(ls, world) != 0) {
            std::fprintf(stderr, "[GEMMUL8 DIST] cannot listen on %s:%d\n", addr, port);
            if (ls >= 0) ::close(ls);
            freeaddrinfo(res);
            return GEMMUL8_E_INTERNAL;
        }
        std::vector<char> served((size_t)world, 0);
        int left = world - 1;
        while (left > 0) {
            const auto now = std::chrono::steady_clock::now();
            if (now >= deadline) break;
            timeval tv{};
            tv.tv_sec = (long)std::chrono::duration_cast<std::chrono::seconds>(deadline - now).count() + 1;
            setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);  // bounds accept()
            const int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) continue;  // timeout or a transient error: the deadline check ends the loop
            set_io_timeout(fd, 5);
            unsigned char salt[16];
            random_salt(salt);
            IdHello h{};
            if (send_all(fd, salt, sizeof salt) && recv_all(fd, &h, sizeof h) && h.magic == kHelloMagic && h.world == (uint32_t)world && h.rank >= 1 &&
                h.rank < (uint32_t)world && h.mac == id_mac(key, salt, h.rank, h.world, (uint32_t)port) && served[h.rank] < 2 && send_all(fd, id, sizeof *id)) {
                if (served[h.rank]++ == 0) --left;
            }
            ::close(fd);
        }
        ::close(ls);
        if (left == 0) rc = 0;
        else std::fprintf(stderr, "[GEMMUL8 DIST] id rendezvous: %d of %d ranks did not connect to %s:%d within %d s\n", left, world - 1, addr, port, limit);
    } else {
        while (rc && std::chrono::steady_clock::now() < deadline) {  // rank 0 may not be listening yet: retry until the deadline
            const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd < 0) break;
            set_io_timeout(fd, 5);
            unsigned char salt[16];
            if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0 && recv_all(fd, salt, sizeof salt)) {
                const IdHello h{kHelloMagic, (uint32_t)rank, (uint32_t)world, 0u, id_mac(key, salt, (uint32_t)rank, (uint32_t)world, (uint32_t)port)};
                if (send_all(fd, &h, sizeof h) && recv_all(fd, id, sizeof *id)) rc = 0;
            }
            ::close(fd);
            if (rc) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (rc) std::fprintf(stderr, "[GEMMUL8 DIST] id rendezvous: rank %d got no id from %s:%d within %d s\n", rank, addr, port, limit);
    }
    freeaddrinfo(res);
    return rc;
}

// ------------------------------------------------------------------------------------------------ default engine: HIP
void* hip_alloc(size_t bytes) {
    void* p = nullptr;
    return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? p : nullptr;
}
void hip_release(void* p) {
    if (p) (void)hipFree(p);
}
int hip_zero(void* p, size_t bytes, void* stream) { return bytes ? (int)hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) : 0; }
int hip_copy(void* dst, const void* src, size_t bytes, void* stream) {
    return bytes ? (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) : 0;
}
int hip_copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, void* stream) {
    return width && height ? (int)hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, (hipStream_t)stream) : 0;
}
const gemmul8_dist_engine kHipEngine = {hip_alloc,           hip_release,          hip_zero,    hip_copy,           hip_copy2d,        gemmul8_scale_bounds,
                                        gemmul8_scale_finish, gemmul8_lowprec_gemm, gemmul8_crt, gemmul8_crt_partial, gemmul8_crt_finish, gemmul8_add_f64};

// ------------------------------------------------------------------------------------------------ partition arithmetic
struct Range {
    size_t b, e;
    size_t size() const { return e - b; }
};
// balanced contiguous split: the first (total % parts) pieces get one extra
Range split_range(size_t total, int parts, int idx) {
    const size_t q = total / (size_t)parts, r = total % (size_t)parts;
    const size_t b = (size_t)idx * q + std::min<size_t>((size_t)idx, r);
    return {b, b + q + ((size_t)idx < r ? 1 : 0)};
}
// Gr x Gc with Gr >= Gc as square as possible.  The row side gets the larger factor: the scaling work of A (row-strided for op N:
// amax pass + LDS-staged extract / quantise) costs more than B's and is divided by Gr.
void block_grid(int world, int* gr, int* gc) {
    int c = 1;
    for (int d = 1; d * d <= world; ++d)
        if (world % d == 0) c = d;
    *gc = c;
    *gr = world / c;
}
size_t elem_bytes(int dtype) { return dtype == GEMMUL8_S ? 4 : dtype == GEMMUL8_Z ? 16 : 8; }
int norm_op(int op) { return (op >= 111 && op <= 113) ? op - 111 : op; }
size_t pad256(size_t x) { return (x + 255) / 256 * 256; }

#define OZ2_RC(expr)            \
    do {                        \
        const int rc__ = (expr); \
        if (rc__) return rc__;  \
    } while (0)

}  // namespace

struct gemmul8_dist_plan {
    gemmul8_comm comm{};
    gemmul8_dist_engine eng{};
    int kind = 0, dtype = 0, backend = 0, opA = 0, opB = 0, fast = 0;
    size_t m = 0, n = 0, k = 0;
    unsigned N = 0;
    int gr = 1, gc = 1;
    Range rows{0, 0}, cols{0, 0};  // the block of C this rank updates
    Range mods{0, 0};              // the moduli it multiplies
    size_t em = 0, en = 0;         // the engine's problem: the block (blocks plan) or the whole matrix (moduli plans)
    bool have = false;             // this rank has engine work
    size_t esz = 8, mid = 1, comps = 1;
    void* work = nullptr;
    gemmul8_layout L{};
    int32_t* mx = nullptr;   // blocks: int32[m + n] bound maxima of the whole problem
    char* recv = nullptr;    // moduli: [N][cols][mp] residue blocks of this rank's columns
    double* part = nullptr;  // fp64sum: [world][hi | lo][cw][mp] partial sums
    double* red = nullptr;   //          [hi | lo][cw][mp] reduced block of this rank
    // fp64sum in moduli groups (GEMMUL8_DIST_FP64_GROUPS = 2 .. 8, round 6): the rank's planes are multiplied group by group; the partial sums of group j
    // go through their own reduce-scatter on the exchange stream while group j + 1 multiplies, and the reduced blocks are added up (hi parts: exact
    // integers, any order; lo parts: one more rounding per group than the single collective -- the same kind of difference the plan already has against
    // one GPU).  Twice the partial-sum buffers, (groups - 1) more passes over the reduced block; what it buys is xGMI time behind matrix time.
    int fgroups = 1;
    double* part2 = nullptr;  // second partial buffer (groups alternate)
    double* red2 = nullptr;   // reduce-scatter target of groups >= 1 before it is added to `red`
    size_t cw = 0, blk = 0;
    void *ev_begin = nullptr, *ev_end = nullptr;  // optional hipEvent_t pair recorded around the low-precision GEMM launch
    void* ev_x[4] = {nullptr, nullptr, nullptr, nullptr};  // optional: around the bounds all-reduce / around the bulk exchange
    void mark(int i, void* stream) const {
        if (ev_x[i] && hip_engine) (void)hipEventRecord((hipEvent_t)ev_x[i], (hipStream_t)stream);
    }
    bool hip_engine = true;
    bool part_clean = false;  // the padding columns of `part` (world * cw > n) are zeroed once, on the first call's stream
    // moduli plan: the rank's planes are multiplied in `groups` groups; the residue exchange of group j runs on a second stream (HIP engine) behind an
    // event while the GEMMs of group j + 1 run on the caller's stream.  GEMMUL8_DIST_GROUPS (default 2; 1 = one exchange behind all GEMMs, the
    // round-4 order) must be the same on every rank: group j of sender s is split_range(its moduli, groups, j) on both sides of every pair.
    int groups = 2;
    bool groups_agreed = false;  // checked across the ranks on the first call (one small all-reduce)
    hipStream_t xstream = nullptr;
    hipEvent_t xev[9] = {};  // [0 .. groups): GEMM group done (caller's stream); [8]: last exchange done (exchange stream)
    char* stage_send = nullptr;  // allgather_c staging (allocated on first use)
    char* stage_recv = nullptr;
    size_t bytes = 0;

    void* alloc(size_t b) {
        void* p = eng.alloc(b);
        if (p) bytes += b;
        return p;
    }
    Range rows_of(int rank) const { return kind == GEMMUL8_DIST_BLOCKS ? split_range(m, gr, rank / gc) : Range{0, m}; }
    Range cols_of(int rank) const {
        if (kind == GEMMUL8_DIST_BLOCKS) return split_range(n, gc, rank % gc);
        if (kind == GEMMUL8_DIST_MODULI) return split_range(n, comm.world, rank);
        const size_t b = std::min(n, (size_t)rank * cw);
        return {b, std::min(n, b + cw)};
    }
};

extern "C" {

int gemmul8_comm_rccl_unique_id(void* id128) {
    if (!id128 || !rccl().ok) return GEMMUL8_E_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return nccl_status(rccl().GetUniqueId(static_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
}

int gemmul8_comm_rccl_create(const void* id128, int rank, int world, gemmul8_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return GEMMUL8_E_ARG;
    if (!rccl().ok) return GEMMUL8_E_UNSUPPORTED;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    RcclCtx* ctx = new (std::nothrow) RcclCtx;
    if (!ctx) return GEMMUL8_E_ARG;
    const int rc = nccl_status(rccl().CommInitRank(&ctx->comm, world, id, rank), "ncclCommInitRank");
    if (rc) {
        delete ctx;
        return rc;
    }
    gemmul8_comm* c = new (std::nothrow) gemmul8_comm{ctx, rank, world, rccl_allreduce_max_i32, rccl_sendrecv, rccl_reduce_scatter_sum_f64, rccl_destroy};
    if (!c) {
        rccl_destroy(ctx);
        return GEMMUL8_E_ARG;
    }
    *out = c;
    return GEMMUL8_OK;
}

int gemmul8_comm_rccl_id_from_env(void* id128, int* rank_out, int* world_out) {
    if (!id128) return GEMMUL8_E_ARG;
    const char* srank = std::getenv("RANK");
    const char* sworld = std::getenv("WORLD_SIZE");
    if (!srank || !sworld) return GEMMUL8_E_ARG;
    const int rank = std::atoi(srank), world = std::atoi(sworld);
    if (world < 1 || rank < 0 || rank >= world) return GEMMUL8_E_ARG;
    if (!rccl().ok) return GEMMUL8_E_UNSUPPORTED;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof id);
    if (rank == 0) OZ2_RC(nccl_status(rccl().GetUniqueId(&id), "ncclGetUniqueId"));
    if (world > 1) {
        const char* addr = std::getenv("MASTER_ADDR");
        const char* sport = std::getenv("GEMMUL8_DIST_PORT");
        int port = sport ? std::atoi(sport) : 0;
        if (!port) {
            const char* mp0 = std::getenv("MASTER_PORT");
            port = (mp0 ? std::atoi(mp0) : 29500) + 17;
        }
        const char* mp = std::getenv("MASTER_PORT");
        const IdKey key = id_key_from_env(mp ? std::atoi(mp) : 29500);
        OZ2_RC(exchange_id_tcp(addr ? addr : "127.0.0.1", port, rank, world, key, &id));
    }
    std::memcpy(id128, &id, sizeof id);
    if (rank_out) *rank_out = rank;
    if (world_out) *world_out = world;
    return GEMMUL8_OK;
}

int gemmul8_comm_rccl_from_env(gemmul8_comm** out) {
    if (!out) return GEMMUL8_E_ARG;
    ncclUniqueId id;
    int rank = 0, world = 1;
    OZ2_RC(gemmul8_comm_rccl_id_from_env(&id, &rank, &world));
    return gemmul8_comm_rccl_create(&id, rank, world, out);
}

int gemmul8_comm_rccl_count(const gemmul8_comm* comm, int* count) {
    if (!comm || !count) return GEMMUL8_E_ARG;
    *count = -1;
    if (comm->destroy != rccl_destroy || !rccl().CommCount) return GEMMUL8_OK;
    return nccl_status(rccl().CommCount(static_cast<RcclCtx*>(comm->ctx)->comm, count), "ncclCommCount");
}

void gemmul8_comm_destroy(gemmul8_comm* comm) {
    if (!comm) return;
    if (comm->destroy) comm->destroy(comm->ctx);
    delete comm;
}

int gemmul8_dist_create(const gemmul8_comm* comm, const gemmul8_dist_engine* engine, int kind, int grid_rows, int dtype, int backend,
                        int op_A, int op_B, size_t m, size_t n, size_t k, unsigned N, int fastmode, gemmul8_dist_plan** out) {
    if (!comm || !out || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world) return GEMMUL8_E_ARG;
    if (kind < GEMMUL8_DIST_BLOCKS || kind > GEMMUL8_DIST_MODULI_FP64SUM) return GEMMUL8_E_ARG;
    if (dtype < 0 || dtype > 3 || backend < 0 || backend > 1) return GEMMUL8_E_ARG;
    if (N < 2 || N > ((dtype == GEMMUL8_S || dtype == GEMMUL8_C) ? 13u : 20u)) return GEMMUL8_E_NUM_MODULI;  // float types: 2..13 (gemmul8.hpp:30)
    op_A = norm_op(op_A), op_B = norm_op(op_B);
    if (op_A < 0 || op_A > 2 || op_B < 0 || op_B > 2 || m == 0 || n == 0 || k == 0) return GEMMUL8_E_ARG;
    if (k > (size_t(1) << 17) || (backend == GEMMUL8_FP8 && k > 65536)) return GEMMUL8_E_ARG;
    if (comm->world > 1 && (!comm->allreduce_max_i32 || !comm->sendrecv || !comm->reduce_scatter_sum_f64)) return GEMMUL8_E_ARG;
    gemmul8_dist_plan* P = new (std::nothrow) gemmul8_dist_plan;
    if (!P) return GEMMUL8_E_ARG;
    P->comm = *comm;
    P->eng = engine ? *engine : kHipEngine;
    P->hip_engine = engine == nullptr;
    if (const char* e = getenv("GEMMUL8_DIST_GROUPS"); e && *e) P->groups = std::max(1, std::min(8, atoi(e)));
    P->kind = kind, P->dtype = dtype, P->backend = backend, P->opA = op_A, P->opB = op_B, P->fast = fastmode ? 1 : 0;
    P->m = m, P->n = n, P->k = k, P->N = N;
    const bool cplx = dtype >= 2;
    P->esz = elem_bytes(dtype);
    P->comps = cplx ? 2 : 1;
    P->mid = (backend == GEMMUL8_INT8 ? 1 : 2) * P->comps;
    const int rank = comm->rank, world = comm->world;
    if (kind == GEMMUL8_DIST_BLOCKS) {
        if (grid_rows > 0) {
            if (world % grid_rows) {
                delete P;
                return GEMMUL8_E_ARG;
            }
            P->gr = grid_rows, P->gc = world / grid_rows;
        } else {
            block_grid(world, &P->gr, &P->gc);
        }
        P->mods = {0, N};
    } else {
        const Range t = split_range(N, world, rank);
        P->mods = t;
        P->cw = (n + (size_t)world - 1) / (size_t)world;
    }
    P->rows = P->rows_of(rank);
    P->cols = P->cols_of(rank);
    if (kind == GEMMUL8_DIST_BLOCKS) {
        P->em = P->rows.size(), P->en = P->cols.size();
        P->have = P->em > 0 && P->en > 0;
    } else {
        P->em = m, P->en = n;
        P->have = true;
    }
    bool ok = true;
    if (P->have) {
        const size_t wbytes = gemmul8_work_size(cplx, backend, P->em, P->en, k, N, 0, 0, nullptr, nullptr);
        P->work = P->alloc(wbytes);
        ok = P->work && gemmul8_get_layout(dtype, backend, P->em, P->en, k, N, P->work, nullptr, nullptr, 0, 0, &P->L) == GEMMUL8_OK;
    }
    if (ok && kind == GEMMUL8_DIST_BLOCKS && !P->fast && world > 1) ok = (P->mx = (int32_t*)P->alloc(4 * (m + n))) != nullptr;
    if (ok && kind == GEMMUL8_DIST_MODULI) ok = (P->recv = (char*)P->alloc(std::max<size_t>(8, (size_t)N * P->cols.size() * P->L.mp * P->mid))) != nullptr;
    if (ok && kind == GEMMUL8_DIST_MODULI_FP64SUM) {
        P->blk = 2 * P->cw * P->L.mp * P->comps;
        P->part = (double*)P->alloc((size_t)world * P->blk * 8);
        P->red = world > 1 ? (double*)P->alloc(P->blk * 8) : nullptr;
        ok = P->part && (world == 1 || P->red);
        if (const char* e = getenv("GEMMUL8_DIST_FP64_GROUPS"); ok && e && *e && world > 1 && P->eng.add_f64)
            P->fgroups = std::max(1, std::min(8, atoi(e)));  // (a rank with fewer planes than groups runs empty groups: the number of collectives is what must agree)
        if (ok && P->fgroups > 1) {
            P->part2 = (double*)P->alloc((size_t)world * P->blk * 8);
            P->red2 = (double*)P->alloc(P->blk * 8);
            ok = P->part2 && P->red2;
        }
    }
    if (ok && ((kind == GEMMUL8_DIST_MODULI && P->groups > 1) || (kind == GEMMUL8_DIST_MODULI_FP64SUM && P->fgroups > 1)) && P->hip_engine && world > 1) {
        // two-stream state of the pipelined exchange: all of it here or none of it (a partial set is released by gemmul8_dist_destroy below)
        ok = hipStreamCreateWithFlags(&P->xstream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < 9; ++i) ok = hipEventCreateWithFlags(&P->xev[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {  // the arguments were valid: what failed is an allocation (or the layout of an allocated workspace)
        std::fprintf(stderr, "[GEMMUL8 DIST] rank %d: plan workspace allocation failed (%zu bytes held so far)\n", rank, P->bytes);
        gemmul8_dist_destroy(P);
        return GEMMUL8_E_INTERNAL;
    }
    *out = P;
    return GEMMUL8_OK;
}

void gemmul8_dist_destroy(gemmul8_dist_plan* P) {
    if (!P) return;
    for (void* p : {(void*)P->work, (void*)P->mx, (void*)P->recv, (void*)P->part, (void*)P->red, (void*)P->part2, (void*)P->red2, (void*)P->stage_send, (void*)P->stage_recv})
        if (p) P->eng.release(p);
    if (P->xstream) (void)hipStreamDestroy(P->xstream);
    for (hipEvent_t e : P->xev)
        if (e) (void)hipEventDestroy(e);
    delete P;
}

size_t gemmul8_dist_workspace_bytes(const gemmul8_dist_plan* P) { return P ? P->bytes : 0; }

int gemmul8_dist_set_events(gemmul8_dist_plan* P, void* ev_begin, void* ev_end) {
    if (!P) return GEMMUL8_E_ARG;
    P->ev_begin = ev_begin, P->ev_end = ev_end;
    return GEMMUL8_OK;
}

int gemmul8_dist_set_exchange_events(gemmul8_dist_plan* P, void* const ev[4]) {
    if (!P) return GEMMUL8_E_ARG;
    for (int i = 0; i < 4; ++i) P->ev_x[i] = ev ? ev[i] : nullptr;
    return GEMMUL8_OK;
}

int gemmul8_dist_exchange_bytes(const gemmul8_dist_plan* P, size_t* allreduce_bytes, size_t* sent, size_t* received) {
    if (!P) return GEMMUL8_E_ARG;
    const int world = P->comm.world, rank = P->comm.rank;
    size_t ar = 0, tx = 0, rx = 0;
    if (world > 1) {
        if (!P->fast) ar = P->kind == GEMMUL8_DIST_BLOCKS ? 4 * (P->m + P->n) : 4 * (P->L.mp + pad256(P->n));
        if (P->kind == GEMMUL8_DIST_MODULI) {
            for (int s = 0; s < world; ++s) {
                if (s == rank) continue;
                tx += P->mods.size() * P->cols_of(s).size() * P->L.mp * P->mid;
                rx += split_range(P->N, world, s).size() * P->cols.size() * P->L.mp * P->mid;
            }
        } else if (P->kind == GEMMUL8_DIST_MODULI_FP64SUM) {
            tx = (size_t)(world - 1) * P->blk * 8;  // every other rank's block of this rank's partial sums
            rx = (size_t)(world - 1) * P->blk * 8;  // the other ranks' partials of this rank's block
        }
    }
    if (allreduce_bytes) *allreduce_bytes = ar;
    if (sent) *sent = tx;
    if (received) *received = rx;
    return GEMMUL8_OK;
}

int gemmul8_dist_owned_block(const gemmul8_dist_plan* P, int rank, size_t* r0, size_t* r1, size_t* c0, size_t* c1) {
    if (!P || rank < 0 || rank >= P->comm.world) return GEMMUL8_E_ARG;
    const Range r = P->rows_of(rank), c = P->cols_of(rank);
    if (r0) *r0 = r.b;
    if (r1) *r1 = r.e;
    if (c0) *c0 = c.b;
    if (c1) *c1 = c.e;
    return GEMMUL8_OK;
}

int gemmul8_dist_my_work(const gemmul8_dist_plan* P, unsigned* moduli, size_t* rows, size_t* cols) {
    if (!P) return GEMMUL8_E_ARG;
    if (moduli) *moduli = (unsigned)P->mods.size();
    if (rows) *rows = P->have ? P->em : 0;
    if (cols) *cols = P->have ? P->en : 0;
    return GEMMUL8_OK;
}

int gemmul8_dist_gemm(gemmul8_dist_plan* P, void* stream, const void* alpha, const void* A, size_t lda, const void* B, size_t ldb,
                      const void* beta, void* C, size_t ldc) {
    if (!P || !alpha || !beta || !A || !B || !C) return GEMMUL8_E_ARG;
    const gemmul8_dist_engine& E = P->eng;
    const gemmul8_comm& X = P->comm;
    const int world = X.world, rank = X.rank;
    const gemmul8_layout* L = &P->L;
    const size_t esz = P->esz, mp = L->mp;
    const unsigned N = P->N;

    if (P->kind == GEMMUL8_DIST_BLOCKS) {
        // op(A) rows [r0, r1): a row block of a column-major m x k matrix (op N) or a column block of the stored k x m one (op T / C)
        const char* As = (const char*)A + (P->opA == 0 ? P->rows.b : P->rows.b * lda) * esz;
        const char* Bs = (const char*)B + (P->opB == 0 ? P->cols.b * ldb : P->cols.b) * esz;
        char* Cs = (char*)C + (P->cols.b * ldc + P->rows.b) * esz;
        if (!P->fast) {
            if (P->have)
                OZ2_RC(E.scale_bounds(stream, P->dtype, P->backend, P->opA, P->opB, P->em, P->en, P->k, As, lda, Bs, ldb, N, 0, P->en, L, 0, 0));
            if (world > 1) {
                // every rank contributes the maxima of its block at the global row / column positions of one zero-filled vector:
                // element-wise MAX over all ranks completes the row maxima (over the column blocks) and the column maxima (over the
                // row blocks) at once
                int32_t* rowmax = (int32_t*)L->scratch;
                int32_t* colmax = rowmax + mp;
                OZ2_RC(E.zero(P->mx, 4 * (P->m + P->n), stream));
                if (P->have) {
                    OZ2_RC(E.copy(P->mx + P->rows.b, rowmax, 4 * P->em, stream));
                    OZ2_RC(E.copy(P->mx + P->m + P->cols.b, colmax, 4 * P->en, stream));
                }
                P->mark(0, stream);
                OZ2_RC(X.allreduce_max_i32(X.ctx, P->mx, P->m + P->n, stream));
                P->mark(1, stream);
                if (P->have) {
                    OZ2_RC(E.copy(rowmax, P->mx + P->rows.b, 4 * P->em, stream));
                    OZ2_RC(E.copy(colmax, P->mx + P->m + P->cols.b, 4 * P->en, stream));
                }
            }
        }
        if (!P->have) return GEMMUL8_OK;
        OZ2_RC(E.scale_finish(stream, P->dtype, P->backend, P->opA, P->opB, P->em, P->en, P->k, As, lda, Bs, ldb, N, P->fast, 0, N, L, 0, 0));
        if (P->ev_begin && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_begin, (hipStream_t)stream);
        OZ2_RC(E.lowprec_gemm(stream, P->dtype, P->backend, P->em, P->en, P->k, N, 0, N, L));
        if (P->ev_end && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_end, (hipStream_t)stream);
        return E.crt(stream, P->dtype, P->backend, N, P->em, P->en, L->C_mid, mp, L->sizeC, L->sftA, L->sftB, alpha, beta, Cs, ldc);
    }

    // ---- moduli-sharded plans: shifts from the whole problem, planes and GEMMs of the rank's moduli only
    const size_t np_ = pad256(P->n);
    const unsigned t0 = (unsigned)P->mods.b, t1 = (unsigned)P->mods.e;
    if (!P->fast) {
        OZ2_RC(E.scale_bounds(stream, P->dtype, P->backend, P->opA, P->opB, P->m, P->n, P->k, A, lda, B, ldb, N, P->cols.b, P->cols.e, L, 0, 0));
        if (world > 1) {
            P->mark(0, stream);
            OZ2_RC(X.allreduce_max_i32(X.ctx, L->scratch, mp + np_, stream));
            P->mark(1, stream);
        }
    }
    OZ2_RC(E.scale_finish(stream, P->dtype, P->backend, P->opA, P->opB, P->m, P->n, P->k, A, lda, B, ldb, N, P->fast, t0, t1, L, 0, 0));
    const size_t ncols = P->cols.size();
    char* Cs = (char*)C + P->cols.b * ldc * esz;
    const char* Cmid = (const char*)L->C_mid;
    const size_t mid = P->mid;

    if (P->kind == GEMMUL8_DIST_MODULI) {
        // Residue exchange, pipelined behind the GEMMs (round 5): the rank's planes go out in `groups` groups; as soon as the GEMMs of group j
        // are done (an event on the caller's stream) its residue blocks travel on the plan's exchange stream -- plane t, columns of rank s ->
        // rank s, into its [t][cols_s][mp] buffer -- while the GEMMs of group j + 1 run.  Both sides of a pair walk a sender's planes in
        // ascending order and cut them into the same groups, so the grouped sends and receives match one to one, call by call.  The CRT (reference
        // order over all N planes of the rank's columns) waits for the last exchange.  A non-HIP engine (the CPU test engine) runs the same
        // sequence on its one stream.
        const size_t slot = ncols * mp * mid;
        const int NG = world > 1 ? P->groups : 1;
        const bool two_streams = P->hip_engine && world > 1 && NG > 1;  // stream + events exist since gemmul8_dist_create
        if (world > 1 && !P->groups_agreed) {
            // GEMMUL8_DIST_GROUPS is read per rank; a mismatch would give unequal grouped send / recv counts per pair = a silent RCCL hang.
            // First call only: all-reduce(MAX) of {groups, -groups} (the first bytes of the receive buffer serve as scratch), one stream
            // synchronisation, and a clear error on every rank instead.
            int32_t v[2] = {P->groups, -P->groups};
            if (P->hip_engine) {
                if (hipMemcpyAsync(P->recv, v, sizeof v, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return GEMMUL8_E_INTERNAL;
            } else {
                std::memcpy(P->recv, v, sizeof v);
            }
            OZ2_RC(X.allreduce_max_i32(X.ctx, P->recv, 2, stream));
            if (P->hip_engine) {
                if (hipMemcpyAsync(v, P->recv, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                    return GEMMUL8_E_INTERNAL;
            } else {
                std::memcpy(v, P->recv, sizeof v);
            }
            if (v[0] != -v[1]) {
                std::fprintf(stderr, "[GEMMUL8 DIST] rank %d: GEMMUL8_DIST_GROUPS differs between ranks (here %d, elsewhere %d ... %d): export the same value to every rank\n",
                             rank, P->groups, -v[1], v[0]);
                return GEMMUL8_E_ARG;
            }
            P->groups_agreed = true;
        }
        void* xs = two_streams ? (void*)P->xstream : stream;
        if (P->ev_begin && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_begin, (hipStream_t)stream);
        for (int j = 0; j < NG; ++j) {
            const Range gj = split_range(P->mods.size(), NG, j);
            const unsigned a = t0 + (unsigned)gj.b, b = t0 + (unsigned)gj.e;
            if (b > a) OZ2_RC(E.lowprec_gemm(stream, P->dtype, P->backend, P->m, P->n, P->k, N, a, b, L));
            if (j == NG - 1 && P->ev_end && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_end, (hipStream_t)stream);
            if (two_streams) {
                if (hipEventRecord(P->xev[j], (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(P->xstream, P->xev[j], 0) != hipSuccess) return GEMMUL8_E_INTERNAL;
            }
            std::vector<gemmul8_p2p_op> ops;
            for (int s = 0; s < world; ++s) {
                const Range sc = P->cols_of(s), st = split_range(N, world, s);
                if (s == rank) {
                    for (unsigned t = a; t < b; ++t) OZ2_RC(E.copy(P->recv + t * slot, Cmid + (t * L->sizeC + sc.b * mp) * mid, slot, xs));
                    continue;
                }
                for (unsigned t = a; t < b && sc.size(); ++t)
                    ops.push_back({(void*)(Cmid + (t * L->sizeC + sc.b * mp) * mid), sc.size() * mp * mid, s, 1});
                const Range sg = split_range(st.size(), NG, j);  // group j of sender s
                for (size_t t = st.b + sg.b; t < st.b + sg.e && ncols; ++t) ops.push_back({P->recv + t * slot, slot, s, 0});
            }
            if (world > 1) {
                if (j == 0) P->mark(2, xs);
                OZ2_RC(X.sendrecv(X.ctx, (int)ops.size(), ops.data(), xs));
                if (j == NG - 1) P->mark(3, xs);
            }
        }
        if (two_streams) {
            if (hipEventRecord(P->xev[8], P->xstream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, P->xev[8], 0) != hipSuccess) return GEMMUL8_E_INTERNAL;
        }
        if (!ncols) return GEMMUL8_OK;
        return E.crt(stream, P->dtype, P->backend, N, P->m, ncols, P->recv, mp, ncols * mp, L->sftA, L->sftB + P->cols.b, alpha, beta, Cs, ldc);
    }

    const size_t half = P->cw * mp * P->comps;  // doubles per (hi | lo) plane of one rank's column block
    if (!P->part_clean) {
        OZ2_RC(E.zero(P->part, (size_t)world * P->blk * 8, stream));
        if (P->part2) OZ2_RC(E.zero(P->part2, (size_t)world * P->blk * 8, stream));
        P->part_clean = true;
    }
    if (P->fgroups > 1) {
        // ---- FP64 partial sums in moduli groups: GEMMs(j) -> partial sums(j) on the caller's stream; reduce-scatter(j) [+ add into the running block]
        // on the exchange stream beside GEMMs(j + 1).  The partial buffers alternate; group j + 2 may only overwrite a buffer once reduce-scatter(j) has
        // read it (xev[4 + (j & 1)] recorded on the exchange stream).  Every rank cuts its own planes into the same NUMBER of groups (a rank
        // with fewer planes than groups contributes zeros in its empty groups: the collective count is what must agree).
        const int NG = P->fgroups;
        const bool two_streams = P->hip_engine && P->xstream != nullptr;
        void* xs = two_streams ? (void*)P->xstream : stream;
        if (!P->groups_agreed) {  // the group count is a collective property (see the moduli plan): one small all-reduce on the first call
            int32_t v[2] = {NG, -NG};
            if (P->hip_engine) {
                if (hipMemcpyAsync(P->red2, v, sizeof v, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return GEMMUL8_E_INTERNAL;
            } else {
                std::memcpy(P->red2, v, sizeof v);
            }
            OZ2_RC(X.allreduce_max_i32(X.ctx, P->red2, 2, stream));
            if (P->hip_engine) {
                if (hipMemcpyAsync(v, P->red2, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                    return GEMMUL8_E_INTERNAL;
            } else {
                std::memcpy(v, P->red2, sizeof v);
            }
            if (v[0] != -v[1]) {
                std::fprintf(stderr, "[GEMMUL8 DIST] rank %d: the fp64sum plan's group count differs between ranks (here %d, elsewhere %d ... %d): export the same "
                                     "GEMMUL8_DIST_FP64_GROUPS to every rank\n", rank, NG, -v[1], v[0]);
                return GEMMUL8_E_ARG;
            }
            P->groups_agreed = true;
        }
        if (P->ev_begin && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_begin, (hipStream_t)stream);
        for (int j = 0; j < NG; ++j) {
            const Range gj = split_range(P->mods.size(), NG, j);
            const unsigned a = t0 + (unsigned)gj.b, b = t0 + (unsigned)gj.e;
            double* pj = (j & 1) ? P->part2 : P->part;
            if (b > a) OZ2_RC(E.lowprec_gemm(stream, P->dtype, P->backend, P->m, P->n, P->k, N, a, b, L));
            if (j == NG - 1 && P->ev_end && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_end, (hipStream_t)stream);
            if (two_streams && j >= 2 && hipStreamWaitEvent((hipStream_t)stream, P->xev[4 + (j & 1)], 0) != hipSuccess) return GEMMUL8_E_INTERNAL;  // buffer free again
            OZ2_RC(E.crt_partial(stream, P->dtype, P->backend, N, a, b, P->m, P->n, Cmid + (size_t)a * L->sizeC * mid, mp, L->sizeC, pj, pj + half, mp, P->cw, P->blk));
            if (two_streams && (hipEventRecord(P->xev[j & 3], (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(P->xstream, P->xev[j & 3], 0) != hipSuccess))
                return GEMMUL8_E_INTERNAL;
            if (j == 0) P->mark(2, xs);
            OZ2_RC(X.reduce_scatter_sum_f64(X.ctx, pj, j == 0 ? P->red : P->red2, P->blk, xs));
            if (two_streams && hipEventRecord(P->xev[4 + (j & 1)], P->xstream) != hipSuccess) return GEMMUL8_E_INTERNAL;
            if (j > 0) OZ2_RC(E.add_f64(xs, P->red, P->red2, P->blk));
            if (j == NG - 1) P->mark(3, xs);
        }
        if (two_streams && (hipEventRecord(P->xev[8], P->xstream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, P->xev[8], 0) != hipSuccess)) return GEMMUL8_E_INTERNAL;
        if (!ncols) return GEMMUL8_OK;
        return E.crt_finish(stream, P->dtype, P->backend, N, P->m, ncols, P->red, P->red + half, mp, L->sftA, L->sftB + P->cols.b, alpha, beta, Cs, ldc);
    }

    if (P->ev_begin && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_begin, (hipStream_t)stream);
    OZ2_RC(E.lowprec_gemm(stream, P->dtype, P->backend, P->m, P->n, P->k, N, t0, t1, L));
    if (P->ev_end && P->hip_engine) (void)hipEventRecord((hipEvent_t)P->ev_end, (hipStream_t)stream);

    // ---- FP64 partial sums + reduce-scatter(sum)
    OZ2_RC(E.crt_partial(stream, P->dtype, P->backend, N, t0, t1, P->m, P->n, Cmid + (size_t)t0 * L->sizeC * mid, mp, L->sizeC, P->part,
                         P->part + half, mp, P->cw, P->blk));
    const double* red = P->part;
    if (world > 1) {
        P->mark(2, stream);
        OZ2_RC(X.reduce_scatter_sum_f64(X.ctx, P->part, P->red, P->blk, stream));
        P->mark(3, stream);
        red = P->red;
    }
    if (!ncols) return GEMMUL8_OK;
    return E.crt_finish(stream, P->dtype, P->backend, N, P->m, ncols, red, red + half, mp, L->sftA, L->sftB + P->cols.b, alpha, beta, Cs, ldc);
}

int gemmul8_dist_allgather_c(gemmul8_dist_plan* P, void* stream, void* C, size_t ldc) {
    if (!P || !C) return GEMMUL8_E_ARG;
    const gemmul8_comm& X = P->comm;
    if (X.world == 1) return GEMMUL8_OK;
    const gemmul8_dist_engine& E = P->eng;
    const size_t esz = P->esz;
    // blocks travel packed (contiguous rows x cols); a block that spans all rows of a C with ldc == m is already packed
    const bool direct = ldc == P->m && P->kind != GEMMUL8_DIST_BLOCKS;
    if (!direct && !P->stage_send) {
        size_t mine = P->rows.size() * P->cols.size() * esz, others = 0;
        for (int s = 0; s < X.world; ++s)
            if (s != X.rank) others += P->rows_of(s).size() * P->cols_of(s).size() * esz;
        P->stage_send = (char*)P->alloc(std::max<size_t>(mine, 1));
        P->stage_recv = (char*)P->alloc(std::max<size_t>(others, 1));
        if (!P->stage_send || !P->stage_recv) return GEMMUL8_E_INTERNAL;
    }
    std::vector<gemmul8_p2p_op> ops;
    const size_t my_bytes = P->rows.size() * P->cols.size() * esz;
    char* Cb = (char*)C;
    void* my_buf = direct ? (void*)(Cb + P->cols.b * ldc * esz) : (void*)P->stage_send;
    if (!direct && my_bytes)
        OZ2_RC(E.copy2d(P->stage_send, P->rows.size() * esz, Cb + (P->cols.b * ldc + P->rows.b) * esz, ldc * esz, P->rows.size() * esz, P->cols.size(), stream));
    size_t off = 0;
    for (int s = 0; s < X.world; ++s) {
        if (s == X.rank) continue;
        const Range r = P->rows_of(s), c = P->cols_of(s);
        const size_t bytes = r.size() * c.size() * esz;
        if (my_bytes) ops.push_back({my_buf, my_bytes, s, 1});
        if (bytes) ops.push_back({direct ? (void*)(Cb + c.b * ldc * esz) : (void*)(P->stage_recv + off), bytes, s, 0});
        off += bytes;
    }
    OZ2_RC(X.sendrecv(X.ctx, (int)ops.size(), ops.data(), stream));
    if (direct) return GEMMUL8_OK;
    off = 0;
    for (int s = 0; s < X.world; ++s) {
        if (s == X.rank) continue;
        const Range r = P->rows_of(s), c = P->cols_of(s);
        const size_t bytes = r.size() * c.size() * esz;
        if (bytes) OZ2_RC(E.copy2d(Cb + (c.b * ldc + r.b) * esz, ldc * esz, P->stage_recv + off, r.size() * esz, r.size() * esz, c.size(), stream));
        off += bytes;
    }
    return GEMMUL8_OK;
}

}  // extern "C"
