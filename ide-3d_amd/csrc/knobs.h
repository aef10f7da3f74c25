// knobs.h — EVERY environment switch of libide3d_hip.so, in one table (round 6: the switches used to be 32 scattered getenv calls, some of
// them read per launch).  None of them is needed in production: each is the "before" of a rule that was measured and kept (the rule's
// comment names the switch), or a fallback a test compares against.
//
//   read ONCE per process (`knobs()`, first use):
//     IDE3D_CONV_ARITH = fp32 | bf16x6 | bf16x3 | f16x3        process default of ide3d_set_conv_arithmetic (include/ide3d_hip.h)
//     planner A/B switches of modconv.hip `plan_conv` (present = the older rule):
//       IDE3D_MODCONV_NO_FLAT  _NO_TCONV3A  _TILE=n  _DEBUG=n  _TA_ROWS=n  _TA_BM64  _HEAD_BM128  _TA_KC8  _NO_SMALLMAP  _TA_OLD  _SP_OLDPLAN
//       _NO_PH32  _NO_W8SPLIT  _NO_ONE_ROUND  _SP_MODES=n  _ALLCLS_MIN=n  _SPLIT_MIN=n  _SP_MINCIN=n  _SP_ROWS=n  _SP_MAXLDS=n  _SPLITK=n
//       _SP_W4  _SP_WBUF2  _HEAD_FP32      IDE3D_SP_W8=n  IDE3D_SP_NO_TEAMS  IDE3D_HEAD_NO_SMALL  IDE3D_HEAD_NO_RESIDENT
//     IDE3D_COMPOSITE_NO_LDS  IDE3D_COMPOSITE_MODE=n           compositing kernel forms (composite.hip)
//     IDE3D_FLR_GENERIC                                        filtered_lrelu: the runtime-parameterised kernel for every shape
//     IDE3D_GATHER_NO_TILE  IDE3D_GATHER_SEGS=n  IDE3D_GATHER_PC=4|8|0   tri-plane gather forms (triplane.hip, triplane_tile.hip)
//     IDE3D_MAPPING_PER_LAYER                                  mapping network as one launch per layer (the form of devices where the one-launch kernel is not co-resident)
//   read PER CALL (`knob_live`): the six fallbacks that tests/ flip inside one process to compare a lean kernel with the form it replaced
//     IDE3D_FIR_NO_LEAN  IDE3D_FIR_NO_CELL  IDE3D_BIAS_ACT_NO_PLANES  IDE3D_MODCONV_NO_STRIP  IDE3D_MODCONV_NO_R16  IDE3D_MODCONV_PAIR=0|1
#pragma once
#include <stdlib.h>
#include <string.h>

namespace ide3d {

struct Knobs {
    int conv_arith;                                   // 1 / 3 / 6 / 16
    bool mc_no_flat, mc_no_allcls; int mc_tile, mc_debug, mc_ta_rows;
    bool ta_bm64, head_bm128, ta_kc8, no_smallmap, ta_old, sp_oldplan, no_ph32, no_w8split, no_one_round; int sp_modes;
    int allcls_min, split_min, sp_min_cin, sp_rows, sp_maxlds, force_split;
    int sp_w8;                                        // -1: not set
    bool sp_no_teams, sp_w4, sp_wbuf2, head_fp32, head_no_small, head_no_resident;
    bool composite_no_lds; int composite_mode;
    bool flr_generic;
    bool gather_no_tile; int gather_segs, gather_pc;
    bool mapping_per_layer;
};

inline const Knobs& knobs() {
    static const Knobs k = [] {
        auto on = [](const char* n) { return getenv(n) != nullptr; };
        auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        Knobs k{};
        const char* a = getenv("IDE3D_CONV_ARITH");
        k.conv_arith = !a ? 6 : (!strcmp(a, "fp32") || !strcmp(a, "1")) ? 1 : (!strcmp(a, "bf16x3") || !strcmp(a, "3")) ? 3
                     : (!strcmp(a, "f16x3") || !strcmp(a, "16")) ? 16 : 6;
        k.mc_no_flat = on("IDE3D_MODCONV_NO_FLAT"); k.mc_no_allcls = on("IDE3D_MODCONV_NO_TCONV3A");
        k.mc_tile = num("IDE3D_MODCONV_TILE", -1); k.mc_debug = num("IDE3D_MODCONV_DEBUG", 0); k.mc_ta_rows = num("IDE3D_MODCONV_TA_ROWS", 0);
        k.ta_bm64 = on("IDE3D_MODCONV_TA_BM64"); k.head_bm128 = on("IDE3D_MODCONV_HEAD_BM128"); k.ta_kc8 = on("IDE3D_MODCONV_TA_KC8");
        k.no_smallmap = on("IDE3D_MODCONV_NO_SMALLMAP"); k.ta_old = on("IDE3D_MODCONV_TA_OLD"); k.sp_oldplan = on("IDE3D_MODCONV_SP_OLDPLAN");
        k.no_ph32 = on("IDE3D_MODCONV_NO_PH32"); k.no_w8split = on("IDE3D_MODCONV_NO_W8SPLIT"); k.no_one_round = on("IDE3D_MODCONV_NO_ONE_ROUND");
        k.sp_modes = num("IDE3D_MODCONV_SP_MODES", 3);
        k.allcls_min = num("IDE3D_MODCONV_ALLCLS_MIN", 4); k.split_min = num("IDE3D_MODCONV_SPLIT_MIN", 512);
        k.sp_min_cin = num("IDE3D_MODCONV_SP_MINCIN", 0); k.sp_rows = num("IDE3D_MODCONV_SP_ROWS", 0); k.sp_maxlds = num("IDE3D_MODCONV_SP_MAXLDS", 0);
        k.force_split = num("IDE3D_MODCONV_SPLITK", 0);
        k.sp_w8 = num("IDE3D_SP_W8", -1);
        k.sp_no_teams = on("IDE3D_SP_NO_TEAMS"); k.sp_w4 = on("IDE3D_MODCONV_SP_W4"); k.sp_wbuf2 = on("IDE3D_MODCONV_SP_WBUF2");
        k.head_fp32 = on("IDE3D_MODCONV_HEAD_FP32"); k.head_no_small = on("IDE3D_HEAD_NO_SMALL"); k.head_no_resident = on("IDE3D_HEAD_NO_RESIDENT");
        k.composite_no_lds = on("IDE3D_COMPOSITE_NO_LDS"); k.composite_mode = num("IDE3D_COMPOSITE_MODE", 0);
        k.flr_generic = on("IDE3D_FLR_GENERIC");
        k.mapping_per_layer = on("IDE3D_MAPPING_PER_LAYER");
        k.gather_no_tile = on("IDE3D_GATHER_NO_TILE"); k.gather_segs = num("IDE3D_GATHER_SEGS", 0); k.gather_pc = num("IDE3D_GATHER_PC", 8);
        return k;
    }();
    return k;
}

// the switches tests flip inside one process: read when asked
inline bool knob_live(const char* name) { return getenv(name) != nullptr; }
inline const char* knob_live_str(const char* name) { return getenv(name); }

}  // namespace ide3d
