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
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
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
