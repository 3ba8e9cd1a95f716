// serl_kregs.h -- a wavefront's register set of f64 literals (round 5).
//
// An f64 literal whose low dword is not zero costs two 32-bit moves at every use; in the one-episode team kernel, whose wavefronts keep one
// role for the whole episode, each wavefront loads the literals of ITS role once -- from its row of an LDS table, which makes them
// wave-uniform VGPR pairs the compiler can neither rematerialise nor move into the (spilling) SGPR file -- and the code reads them through
// CITW_K (generated glue, citation_wave.h), CITW_LK (citation_libm.h) and DET_K (rollout_device.h).  Each of those is `HAVE_K ? KR.k[slot] :
// literal` with HAVE_K a compile-time fact after inlining: callers that pass no set compile the literals, as before.
#pragma once
#ifndef CITW_NK
#define CITW_NK 64
#endif
struct CitwKRegs { double k[CITW_NK]; };
static __device__ const CitwKRegs citw_no_kregs = {};      // what the default arguments bind when there is no set (never read: HAVE_K is false there)
