// prof.h -- profiling scope: brackets the launches of one group with HIP events (nvqa_profile_*)
#pragma once
#include "nvqa_ctx.h"

namespace nvqa {
struct ProfScope {
    nvqa_ctx *c;
    int id;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(nvqa_ctx *c_, int id_, double flops = 0, double bytes = 0, hipStream_t st_ = nullptr)
        : c(c_), id(id_), st(st_ ? st_ : c_->s)
    {
        if (!c->prof_on) return;
        // events come from a pool (prof_collect returns them): creating two per scope made the host the bottleneck of a profiled
        // step -- the queue ran empty between launches and the first profiled steps measured 5-10 % long
        auto take = [&]() { hipEvent_t e = nullptr; if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
        e0 = take();
        e1 = take();
        (void)hipEventRecord(e0, st);
        c->prof[id].flops += flops;
        c->prof[id].bytes += bytes;
        c->prof[id].launches += 1;
    }
    ~ProfScope()
    {
        if (!c->prof_on) return;
        (void)hipEventRecord(e1, st);
        c->prof[id].pending.emplace_back(e0, e1);
    }
};

} // namespace nvqa
