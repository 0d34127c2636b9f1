// fused_geom.hpp - the LDS geometry of the static fused programs as COMPILE-TIME constants.
//
// Why (round 4).  fp32 MFMA and VALU work share a SIMD's issue port (profiles/r03_valu_next_to_mfma.txt), so every
// instruction of an op's tails is paid in matrix-pipe time at large batch, and at B = 100 the tails are chains of
// s_load -> s_waitcnt -> address -> ds_read.  The ISA census (tools/isa_census.py, profiles/r04_isa_census_*.txt) of the round-3
// programs showed ~65 address VALU instructions + ~90 SALU / 11 s_load per op that exist only because the buffer offsets and row
// strides of an op were RUNTIME descriptor fields, and a prologue of 1 177 VALU + 492 SALU + 151 branches (generic staging loops,
// divisions by runtime widths).  The host's buffer placement for the standard networks never changes, so it is written down here;
// every LDS address of a static program becomes `lane base + immediate`.
//
// Safety.  The host still COMPUTES the geometry (build_fused_segment, mpdx.hip: live-range placement, bank-conflict-searched row
// strides) and compares every field with the table of the program it matched (fused_geom_matches); on any difference the segment
// runs on the generic op-list kernel with runtime descriptors - slower, never wrong (tests/test_abi_cpu.py asserts that the
// standard networks do get the static programs).  To regenerate a table: MPDX_DEBUG_FUSE=2 prints it in this initialiser form.
#pragma once

namespace mpdx {

constexpr int kGeomMaxOps = 16;

struct FusedGeomOp { int shape, src_off4, src_rs4, rsrc_off4, rsrc_rs4, res_off4, res_rs4, dst_off4, dst_rs4, gdst, p_off, tb_off; };
struct FusedGeom {
    int nops;
    int in_off4, in_rs4, in_rows, L0;
    int gc1, gc2;             // channels of the staged input; gc1 == -1: the network input [B][L0][state_dim], state_dim read at run time (<= 16)
    int c3, L3, s3_off4, s3_rs4, s3_col4;
    int stat_off, par_off, par_floats, tt_off, tt_n, fpar_off;   // par_floats == -1: depends on state_dim (final_conv[1]'s weights), read at run time
    int H, Cf;                // final op (0: none)
    FusedGeomOp ops[kGeomMaxOps];
};

struct GeomNone { static constexpr bool has = false; static constexpr FusedGeom g = {}; };   // runtime geometry (g unused)

// downs.0 + downs.1 + downs.2 of dim_mults (1, 2, 4, 8): program 5 (FusedSeqDown3)
struct GeomDown3 {
    static constexpr bool has = true;
    static constexpr FusedGeom g = { 15, 0, 6, 68, 64, -1, 0, 0, 0, 0, 0, 0, 9792, 10304, 4480, 9856, 448, 0, 0, 0, {
        {0, 0, 6, 0, 0, -1, 0, 408, 10, -1, 0, 0},
        {1, 408, 10, 0, 6, -1, 0, 1088, 10, -1, 128, -1},
        {2, 1088, 10, 0, 0, -1, 0, 408, 10, -1, 256, 32},
        {2, 408, 10, 0, 0, 1088, 10, 1768, 10, -1, 384, -1},
        {3, 1768, 10, 0, 0, -1, 0, 0, 10, -1, 512, -1},
        {4, 0, 10, 0, 0, -1, 0, 360, 18, -1, 640, 64},
        {5, 360, 18, 0, 10, -1, 0, 1008, 18, -1, 896, -1},
        {6, 1008, 18, 0, 0, -1, 0, 360, 18, -1, 1152, 128},
        {6, 360, 18, 0, 0, 1008, 18, 1656, 18, 0, 1408, -1},
        {7, 1656, 18, 0, 0, -1, 0, 0, 18, -1, 1664, -1},
        {16, 0, 18, 0, 0, -1, 0, 360, 34, -1, 1920, 192},
        {17, 360, 34, 0, 18, -1, 0, 1040, 34, -1, 2432, -1},
        {18, 1040, 34, 0, 0, -1, 0, 360, 34, -1, 2944, 320},
        {18, 360, 34, 0, 0, 1040, 34, 1720, 34, 1, 3456, -1},
        {19, 1720, 34, 0, 0, -1, 0, -1, 0, 2, 3968, -1},
    }};
};

// downs.0 + downs.1 of dim_mults (1, 2, 4): program 0 (FusedSeqDown)
struct GeomDown {
    static constexpr bool has = true;
    static constexpr FusedGeom g = { 10, 0, 6, 68, 64, -1, 0, 0, 0, 0, 0, 0, 9792, 10048, 1920, 9856, 192, 0, 0, 0, {
        {0, 0, 6, 0, 0, -1, 0, 408, 10, -1, 0, 0},
        {1, 408, 10, 0, 6, -1, 0, 1088, 10, -1, 128, -1},
        {2, 1088, 10, 0, 0, -1, 0, 408, 10, -1, 256, 32},
        {2, 408, 10, 0, 0, 1088, 10, 1768, 10, -1, 384, -1},
        {3, 1768, 10, 0, 0, -1, 0, 0, 10, -1, 512, -1},
        {4, 0, 10, 0, 0, -1, 0, 360, 18, -1, 640, 64},
        {5, 360, 18, 0, 10, -1, 0, 1008, 18, -1, 896, -1},
        {6, 1008, 18, 0, 0, -1, 0, 360, 18, -1, 1152, 128},
        {6, 360, 18, 0, 0, 1008, 18, 1656, 18, 0, 1408, -1},
        {7, 1656, 18, 0, 0, -1, 0, -1, 0, 1, 1664, -1},
    }};
};

// downs.2 (no Downsample1d) + mid_block1 + mid_block2 of dim_mults (1, 2, 4) - eight Conv1dBlocks of 128 channels on 16 positions: program 6 (FusedSeqMid3)
struct GeomMid3 {
    static constexpr bool has = true;
    static constexpr FusedGeom g = { 8, 0, 18, 20, 16, 64, 0, 0, 0, 0, 0, 0, 9600, 10176, 4096, 9664, 512, 0, 0, 0, {
        {16, 0, 18, 0, 0, -1, 0, 360, 34, -1, 0, 0},
        {17, 360, 34, 0, 18, -1, 0, 1040, 34, -1, 512, -1},
        {18, 1040, 34, 0, 0, -1, 0, 360, 34, -1, 1024, 128},
        {18, 360, 34, 0, 0, 1040, 34, 1720, 34, 0, 1536, -1},
        {18, 1720, 34, 0, 0, -1, 0, 360, 34, -1, 2048, 256},
        {18, 360, 34, 0, 0, 1720, 34, 1040, 34, -1, 2560, -1},
        {18, 1040, 34, 0, 0, -1, 0, 360, 34, -1, 3072, 384},
        {18, 360, 34, 0, 0, 1040, 34, -1, 0, 1, 3584, -1},
    }};
};

// the last two up levels + final_conv + DDPM step (both standard networks): program 3 (FusedSeqUpAB)
struct GeomUpAB {
    static constexpr bool has = true;
    static constexpr FusedGeom g = { 12, 0, 66, 20, 16, 128, 128, 64, 32, 1320, 34, 16, 13056, 13312, -1, 13120, 192, 2048, 64, 32, {
        {8, 0, 66, 0, 0, -1, 0, 2544, 18, -1, 0, 0},
        {9, 2544, 18, 0, 66, -1, 0, 2904, 18, -1, 256, -1},
        {10, 2904, 18, 0, 0, -1, 0, 2544, 18, -1, 512, 64},
        {10, 2544, 18, 0, 0, 2904, 18, 0, 66, -1, 768, -1},
        {11, 0, 66, 0, 0, -1, 0, 1320, 34, -1, 1024, -1},
        {12, 1320, 34, 0, 0, -1, 0, 0, 10, -1, 1280, 128},
        {13, 0, 10, 1320, 34, -1, 0, 360, 10, -1, 1408, -1},
        {14, 360, 10, 0, 0, -1, 0, 0, 10, -1, 1536, 160},
        {14, 0, 10, 0, 0, 360, 10, 1320, 34, -1, 1664, -1},
        {15, 1320, 34, 0, 0, -1, 0, 0, 10, -1, 1792, -1},
        {2, 0, 10, 0, 0, -1, 0, 680, 10, -1, 1920, -1},
        {63, 680, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    }};
};

}  // namespace mpdx
