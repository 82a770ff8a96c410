// arena.h — the workspace allocator both schedules use (engine.hip: fp16 U-Net / VAE / CLIP; unet_f32.hip: the fp32 U-Net).
// A first-fit free list over ONE device allocation, driven from the host: the schedule is run once "dry" (sizes only, no
// launches) to learn its exact peak, the buffer is (re)allocated only when a call needs more than it holds, and the real run
// then hands out the same offsets again (the allocator is deterministic).
#pragma once
#include <stddef.h>
#include <vector>

namespace dm {

struct Arena {
    struct Blk { size_t off, sz; bool free; };
    std::vector<Blk> blks;
    char* base = nullptr;
    size_t cap = 0, peak = 0;
    bool dry = false;
    void reset(size_t capacity, bool dry_run) {
        blks.clear(); blks.push_back({0, capacity, true}); peak = 0; dry = dry_run;
    }
    // returns offset or (size_t)-1
    size_t alloc(size_t n) {
        n = (n + 255) & ~(size_t)255;
        if (n == 0) n = 256;
        for (size_t i = 0; i < blks.size(); ++i) {
            if (blks[i].free && blks[i].sz >= n) {
                const size_t off = blks[i].off;
                if (blks[i].sz > n) {
                    Blk rest{off + n, blks[i].sz - n, true};
                    blks[i].sz = n; blks[i].free = false;
                    blks.insert(blks.begin() + i + 1, rest);
                } else blks[i].free = false;
                if (off + n > peak) peak = off + n;
                return off;
            }
        }
        return (size_t)-1;
    }
    void release(size_t off) {
        for (size_t i = 0; i < blks.size(); ++i) {
            if (blks[i].off == off && !blks[i].free) {
                blks[i].free = true;
                if (i + 1 < blks.size() && blks[i + 1].free) { blks[i].sz += blks[i + 1].sz; blks.erase(blks.begin() + i + 1); }
                if (i > 0 && blks[i - 1].free) { blks[i - 1].sz += blks[i].sz; blks.erase(blks.begin() + i); }
                return;
            }
        }
    }
};

}  // namespace dm
