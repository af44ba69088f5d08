// Device-memory arena of one zk_ctx: grow-only hipMalloc slabs with a best-fit, coalescing free list.
//
// Why not hipMallocAsync: on this ROCm (7.2) a 40 GB request costs 0.2-2.7 s whether it is a fresh hipMalloc
// or a "reuse" from the stream-ordered pool (measured: tools/scratch/pooltest.hip, profiles/archive/r01f_pool_alloc.txt),
// which at 2^20 rows is more than the whole proof.  A table commitment at that size needs 20-40 GB buffers
// (Keccak: 2431 columns -> 40.8 GB of LDE), so the library keeps the HBM it has touched and hands it out again
// in O(log blocks) host time.  288 GB of HBM is the budget this is sized for: nothing is returned to the driver
// until zk_ctx_mem_trim / zk_ctx_destroy, or until a hipMalloc fails (then free slabs are released and the
// request retried).
//
// Ordering contract: every user of the arena enqueues its work on ctx->stream, so a block freed on the host
// may be handed out again immediately -- later kernels are ordered behind earlier ones by the stream.
// zk_ctx_set_stream drains the old stream before switching.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <vector>

struct DevArena {
    static constexpr size_t ALIGN = 512;
    static constexpr size_t SLAB_ALIGN = size_t(2) << 20;
    static constexpr size_t MIN_SLAB = size_t(256) << 20;
    struct Slab { char *base; size_t size; };
    std::vector<Slab> slabs;
    std::map<char *, size_t> free_by_addr;             // start -> size
    std::multimap<size_t, char *> free_by_size;        // size -> start
    std::unordered_map<void *, size_t> live;           // handed-out blocks
    size_t reserved = 0, in_use = 0, peak_in_use = 0;

    static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

    void erase_size_entry(size_t size, char *p) {
        auto r = free_by_size.equal_range(size);
        for (auto it = r.first; it != r.second; ++it)
            if (it->second == p) { free_by_size.erase(it); return; }
    }
    void insert_free(char *p, size_t size) {
        free_by_addr[p] = size;
        free_by_size.emplace(size, p);
    }
    const Slab *slab_of(const char *p) const {
        for (const Slab &s : slabs)
            if (p >= s.base && p < s.base + s.size) return &s;
        return nullptr;
    }

    hipError_t grow(size_t bytes) {
        size_t sz = round_up(bytes < MIN_SLAB ? MIN_SLAB : bytes, SLAB_ALIGN);
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess && sz > bytes) {           // try the exact size before giving up
            (void)hipGetLastError();
            sz = round_up(bytes, SLAB_ALIGN);
            e = hipMalloc(&p, sz);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
        slabs.push_back({(char *)p, sz});
        reserved += sz;
        insert_free((char *)p, sz);
        return hipSuccess;
    }

    hipError_t alloc(void **out, size_t bytes) {
        *out = nullptr;
        size_t need = round_up(bytes ? bytes : 1, ALIGN);
        auto it = free_by_size.lower_bound(need);
        if (it == free_by_size.end()) {
            hipError_t e = grow(need);
            if (e != hipSuccess) {                     // fragmentation or genuine OOM: release idle slabs, retry
                trim();
                e = grow(need);
                if (e != hipSuccess) return e;
            }
            it = free_by_size.lower_bound(need);
        }
        char *p = it->second;
        size_t have = it->first;
        free_by_size.erase(it);
        free_by_addr.erase(p);
        if (have > need) insert_free(p + need, have - need);
        live[p] = need;
        in_use += need;
        if (in_use > peak_in_use) peak_in_use = in_use;
        *out = p;
        return hipSuccess;
    }

    void free(void *ptr) {
        if (!ptr) return;
        auto lt = live.find(ptr);
        if (lt == live.end()) return;                  // not ours (or double free): ignore
        char *p = (char *)ptr;
        size_t size = lt->second;
        live.erase(lt);
        in_use -= size;
        const Slab *s = slab_of(p);
        // coalesce with the free neighbours inside the same slab
        auto nx = free_by_addr.find(p + size);
        if (nx != free_by_addr.end() && s && p + size < s->base + s->size) {
            erase_size_entry(nx->second, nx->first);
            size += nx->second;
            free_by_addr.erase(nx);
        }
        auto pv = free_by_addr.lower_bound(p);
        if (pv != free_by_addr.begin()) {
            --pv;
            if (pv->first + pv->second == p && s && pv->first >= s->base) {
                erase_size_entry(pv->second, pv->first);
                p = pv->first;
                size += pv->second;
                free_by_addr.erase(pv);
            }
        }
        insert_free(p, size);
    }

    // hipFree every slab that is entirely free; returns the bytes released
    size_t trim() {
        size_t released = 0;
        for (size_t i = 0; i < slabs.size();) {
            auto f = free_by_addr.find(slabs[i].base);
            if (f != free_by_addr.end() && f->second == slabs[i].size) {
                erase_size_entry(f->second, f->first);
                free_by_addr.erase(f);
                (void)hipFree(slabs[i].base);
                released += slabs[i].size;
                reserved -= slabs[i].size;
                slabs[i] = slabs.back();
                slabs.pop_back();
            } else {
                ++i;
            }
        }
        return released;
    }

    // make sure one free block of at least `bytes` exists
    hipError_t reserve(size_t bytes) {
        size_t need = round_up(bytes, ALIGN);
        if (free_by_size.lower_bound(need) != free_by_size.end()) return hipSuccess;
        return grow(need);
    }

    void destroy() {
        for (Slab &s : slabs) (void)hipFree(s.base);
        slabs.clear(); free_by_addr.clear(); free_by_size.clear(); live.clear();
        reserved = in_use = 0;
    }
};

// The arenas of one zk_ctx.  Lane 0 belongs to ctx->stream; lane 1 to the ctx's side stream (segment_host.inc: the
// auxiliary-commitment pipeline that runs beside the per-table chain).  Each lane recycles blocks in the order of ITS
// stream, so a block never migrates between streams without a host-side synchronisation in between: allocation takes
// from the current lane, a free returns the block to the lane that owns it.
struct ArenaSet {
    DevArena lane[2];
    int cur = 0;
    // A lane's own out-of-memory recovery (DevArena::alloc: trim, grow again) only releases ITS idle slabs.  When that
    // still fails, the idle slabs of the OTHER lane go back to the driver as well and the request is retried: a side-lane
    // allocation must not fail while tens of GB sit unused in lane 0 (r03 advisor, medium).  hipFree synchronises the
    // device, and only wholly free slabs are released, so no kernel in flight can be touching them.
    hipError_t alloc(void **out, size_t bytes) {
        hipError_t e = lane[cur].alloc(out, bytes);
        if (e != hipSuccess && lane[cur ^ 1].trim() > 0) e = lane[cur].alloc(out, bytes);
        return e;
    }
    void free(void *p) {
        if (!p) return;
        if (lane[0].live.count(p)) lane[0].free(p);
        else lane[1].free(p);
    }
    hipError_t reserve(size_t bytes) {
        hipError_t e = lane[0].reserve(bytes);
        if (e != hipSuccess && lane[1].trim() > 0) e = lane[0].reserve(bytes);
        return e;
    }
    size_t trim() { return lane[0].trim() + lane[1].trim(); }
    void destroy() { lane[0].destroy(); lane[1].destroy(); }
    size_t reserved() const { return lane[0].reserved + lane[1].reserved; }
    size_t in_use() const { return lane[0].in_use + lane[1].in_use; }
    size_t peak_in_use() const { return lane[0].peak_in_use + lane[1].peak_in_use; }
};
