// Launch-plan options of libmmdeer_hip.so (include/mmdeer.h: mmdeer_set_option / mmdeer_get_option).
// The library reads no environment variable: the defaults are the shipped plan, and a host that wants another one says
// so through the C ABI (mmdeer/_lib.py forwards MMDEER_<NAME> environment variables at load time for the A/B tools).
// Options are read at every call, so a process can switch plans between calls; the table is the library's only mutable
// process-wide state besides the communicator handles of comm.hip.
#pragma once

namespace mmdeer {

enum OptId {
  OPT_FUSED_ATTN = 0,   // 1: tri_fused.hip (in_proj + attention in one kernel, bf16 mode); 0: unfused pair
  OPT_QKV_RECOMPUTE,    // 1: the fused backward recomputes the head tiles; 0: the forward also stores q|k|v
  OPT_XCD,              // 1: XCD-contiguous workgroup renumbering in the GEMM kernels
  OPT_NT128,            // 1: one 8-wave 128x128 LDS-DMA workgroup per CU where 128x64 tiles would need two
  OPT_NT192,            // 1: 256x192 tiles in the 256-row forward kernel when they fill the chip in one round
  OPT_GLDS,             // 1: LDS-DMA GEMM kernels; 0: register-staged kernels everywhere
  OPT_NT8,              // 1: the 8-wave 128x64 form of the LDS-DMA kernel
  OPT_T128,             // smallest 128x64 tile count that selects the 128x64 kernel
  OPT_TILE,             // -1: automatic; 0..3: force a GemmTile
  OPT_KSTEPS,           // 0: automatic; > 0: K-tiles per split-K slice of a weight-gradient problem
  OPT_LN_FUSED,         // 1: every LayerNorm of the forward runs inside the GEMM that consumes it (gemm_ln.hip, bf16 mode)
  OPT_CHAIN,            // 1: the row-local layer chains of the forward run as single launches (chain.hip, bf16 mode)
  OPT_CHAIN_BWD,        // 1 (with chain = 1): the head / trimodal dX products and LayerNorm backwards of the backward pass as one chain launch
  OPT_CHAIN_MIN,        // smallest batch that takes the chains (default 512: measured wins down to there; tests lower it)
  OPT_DW_TILE,          // GemmTile of the weight-gradient launch: 2 = 128x128 tiles without split-K (default), 3 = 256x256 + split-K slabs, 4 = 256x128
  OPT_DW_KG,            // 128x128 weight-gradient tiles: 2 = the workgroup's halves split each 64-row stage of K (default), 1 = 32-row stages
  OPT_CHAIN_MAX,        // largest batch that takes the chains
  OPT_CHAIN_NIG,        // 1 (with the backward chain, loss mode): the head's last-layer backward + loss gradient run in the chain's prologue
  OPT_SPLITK_MAX,       // largest number of split-K slices of a weight-gradient problem (slabs: 4 B per parameter per slice)
  OPT_CHAIN_DEPTH,      // weight stages a wave of the 16-sample layer-chain kernel keeps in flight: 4 (default), 2, or 8 (two granules of four
                        // slots: parity-green, measured slower -- register spills at 192 compiler-visible registers)
  OPT_CHAIN_TS,         // 0: 16-sample chain workgroups up to B = 4096, 32-sample ones above; 16 / 32: that size at every batch
  OPT_CHAIN_IN,         // 1 (with chain = 1, bf16 feature blocks, B <= 4096): the three input projections and the audio padding run
                        // inside the audio-visual chain's launch instead of as pad + F1 launches
  OPT_CHAIN_NIGF,       // 1 (with chain = 1; default 0): the NIG head (last layer, activations, loss statistics) runs as the tail of the
                        // forward head chain instead of a launch of its own.  Bit-identical, one launch fewer -- and measured 6 us
                        // SLOWER per step at B = 4096: the 256 wave partials cost the backward chain's prologue 8k cycles more to
                        // fetch than the 64 block partials, the tail itself 5k
  OPT_ADAM_FUSED,       // 1 (bf16 mode): mmdeer_adamw_step writes every derived weight image from the update itself, tile by tile (optim.h);
                        // 0: element-wise update + one repack launch
  OPT_COUNT
};

int opt(OptId id);                          // current value
int opt_set(const char* name, int value);   // 0 = ok, -1 = unknown name, -2 = value outside the option's range
int opt_range(const char* name, int* lo, int* hi);
int opt_get(const char* name, int* value);  // 0 = ok, -1 = unknown name
const char* opt_name(int i);                // nullptr past the end

}  // namespace mmdeer
