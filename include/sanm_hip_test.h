/* sanm_hip_test.h -- TEST HOOKS of libsanm_hip.so.
 *
 * Not part of the interface a maintainer of the reference binds (include/sanm_hip.h): these entry points exist for
 * this repository's own tests (fault injection, a compile check of the generated kernels that needs no GPU, cache
 * counters).  They are exported from the same library so that the tests exercise the product's own code paths. */
#ifndef SANM_HIP_TEST_H
#define SANM_HIP_TEST_H
#include "sanm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test hook (fault injection): during the next expansion, corrupt one entry -- kind 1: coefficient x_order[index],
 * 2: right-hand side b_order[index] before its solve, 3: Jacobian values [index, index + max(order, 1)) after the
 * assembly; the entry is multiplied by `value` if scale != 0, else replaced by it (NaN allowed).  The checks the reference makes per
 * order (libsanm/anm.cpp:271-285, sparse_solver.cpp:160-161, :288-289) are batched after the order loop here;
 * tests/test_fault_injection.py uses this hook to show that each of them fires. */
int sanm_anm_debug_inject(sanm_anm_solver* s, int kind, int order, int64_t index, double value, int scale);

/* test hook, needs no device: 0 if `source` (which may include "program.h" / "tet_ops.h") compiles for gfx950
 * with the run-time compiler; the compiler's log goes to `log`, the size of the code object to *code_size. */
int sanm_rtc_compile_check(const char* source, char* log, size_t log_cap, size_t* code_size);
/* how the code objects of the run-time compiled pass kernels were obtained in this process so far: compilations,
 * hits of the in-process cache, hits of the on-disk cache (rtc.cpp) */
int sanm_rtc_cache_stats(int64_t* compiled, int64_t* memory_hits, int64_t* disk_hits);
/* Code objects built ahead of time (sanm_amd/build.py).  The source of the pass kernels depends on the structure of
 * the graph and on the order only -- not on the mesh or the material --, so the build compiles the sources of the fea
 * models' own graphs (fea/material.cpp:20-115 through fea/mesh_template.h:174-219) at the usual orders and embeds the
 * code objects in the library; a solver whose generated source matches one takes it from there (setup profile:
 * "jit_embedded").  sanm_fea_spec_source: that source, from a one-cell mesh, no device needed;
 * sanm_rtc_source_key: the key a source is looked up under (33 bytes incl. the terminator);
 * sanm_rtc_compile_to_file: compile without any cache and write the code object; sanm_rtc_embedded_hits: how many
 * solvers of this process were served from the embedded set. */
int64_t sanm_fea_spec_source(int energy_model, int inverse, int order, char* buf, int64_t cap);
int sanm_rtc_source_key(const char* source, char* key33);
int sanm_rtc_compile_to_file(const char* source, const char* path, char* log, size_t log_cap);
int sanm_rtc_embedded_hits(int64_t* hits);
/* forget the in-process cache of code objects (the on-disk cache stays): the next solver for a known graph loads its
 * kernels from the disk like a fresh process would (bench.py measures that as end_to_end.setup_seconds.jit_cached) */
int sanm_rtc_cache_drop_memory(void);
/* obtains the code object of `source` through the caches exactly like a solver under construction does; 0 = ok */
int sanm_rtc_cache_probe(const char* source);

/* the tree-to-ranks plan of a direct solver created with SANM_MF_PLAN_WORLD=G in the environment (analysis as rank 0
 * of G; scripts/dist_plan.py).  out[0..11] = {world G, stages S, total flops, flops of the top, subtrees, doubles of all
 * Schur exchanges, doubles of all inbox exchanges, factor entries of the top, factor entries in all, critical-path
 * flops, subtree imbalance, 0}; then S * G flops (rank r in stage s at [s * G + r]), S * G factor entries, and per stage
 * the doubles of its Schur exchange and the largest number of them one rank receives, then the number X of Schur
 * transfers and {stage, src, dst, doubles, stage that produces it} for each.  Returns the number of doubles the plan has
 * (12 + 2 S G + 2 S + 1 + 5 X) through *n_out; writes at most cap of them. */
int sanm_direct_solver_dist_plan(const sanm_direct_solver* s, int64_t cap, double* out, int64_t* n_out);

/* Point-to-point transfers of the distributed direct solver through a callback (tests: several ranks of the host harness
 * over gloo run the branch that ncclSend / ncclRecv / ncclBroadcast serve on the library's own communicator).  A transfer
 * moves doubles [off, off + cnt) of `base` (memory the callback can address: the harness's) from rank src to rank dst,
 * the same range on both; dst < 0: from src to everyone.  Every rank is handed the same list; the callback returns when
 * this rank's part is done, 0 = ok.  Solvers created while a callback is set use it for the exchanges between the
 * stages (the all-reduce of the shard description still sums b_k, f(x0), the Jacobian values and the pivot status);
 * fn == NULL: unset. */
typedef struct sanm_test_xfer {
    int32_t src, dst;
    int64_t off, cnt;
    int32_t src_stage;
} sanm_test_xfer;
typedef int (*sanm_test_p2p_fn)(void* user, double* base, const sanm_test_xfer* xfers, int n);
int sanm_test_set_p2p(sanm_test_p2p_fn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* SANM_HIP_TEST_H */
