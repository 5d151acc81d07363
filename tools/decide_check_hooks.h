// decide_check_hooks.h — verification hooks for csrc/render_kernels.hip (NOT part of the product build).
// tools/decide_check.sh compiles a variant of the library with
//     -DMCRT_KERNEL_HOOKS='"decide_check_hooks.h"' -Itools
// in which every record whose whole bundle of shadow rays `lit` has DECIDED (rt::bundle_classify: lit by all S light samples
// or by none) is traced ray by ray as well: the traced count must equal the decided one.  Counters 2000.. of the pass's
// counter array: undecided, decided all-shadowed, decided all-lit, CONTRADICTED; contradictions are printed with the hit.
#ifndef MCRT_DECIDE_CHECK_HOOKS_H
#define MCRT_DECIDE_CHECK_HOOKS_H

// lit_kernel: shared state
#define MCRT_HOOK_LIT_SHARED __shared__ uint32_t s_stat[kBlock];

// lit_kernel, phase A1, behind bundle_classify: remember the decision, then treat the record as undecided with the full
// candidate mask so that phases A2 / B trace all of its rays
#define MCRT_HOOK_LIT_CLASSIFIED(known, undecided, cand, O)       \
    s_stat[threadIdx.x] = static_cast<uint32_t>((known) + 1);     \
    (undecided) = true;                                           \
    (cand) = bundle_candidates<kPosed>(scg, (O), lpos, lradius);

// lit_kernel, phase C: compare
#define MCRT_HOOK_LIT_SHADED(lit, r)                                                                                                   \
    if (mode == SHADOW_SOFT) {                                                                                                         \
        const uint32_t st = s_stat[threadIdx.x];                                                                                       \
        atomicAdd(&ws.counters[2000 + (st == 0u ? 0 : (st == 1u ? 1 : 2))], 1u);                                                       \
        if (st != 0u && st - 1u != (lit)) {                                                                                            \
            atomicAdd(&ws.counters[2003], 1u);                                                                                         \
            printf("DECIDE_CHECK decided %u traced %u  P %.9g %.9g %.9g N %g %g %g depth %d\n", st - 1u, (lit), (r).p.x, (r).p.y, (r).p.z, \
                   (r).n.x, (r).n.y, (r).n.z, (r).depth);                                                                              \
        }                                                                                                                              \
    }

// resolve_kernel: the pass's summary line (tools/decide_check.sh sums them)
#define MCRT_HOOK_RESOLVE_BEGIN()                                                                                                        \
    if (blockIdx.x == 0) { /* before this block moves the counters' base */                                                             \
        if (threadIdx.x == 0)                                                                                                           \
            printf("DECIDE_CHECK records: %u undecided, %u decided all-shadowed, %u decided all-lit, %u CONTRADICTED\n",                 \
                   ws.counters[2000] - ws.counter_base[2000], ws.counters[2001] - ws.counter_base[2001],                                \
                   ws.counters[2002] - ws.counter_base[2002], ws.counters[2003] - ws.counter_base[2003]);                               \
        __syncthreads();                                                                                                                \
    }

#endif
