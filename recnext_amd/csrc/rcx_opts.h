// The RCX_* environment switches (A/B measurement knobs; none changes results beyond the stated tolerances).  They are read ONCE, at the
// first call that asks for one, into process-wide storage -- a launch no longer pays ~35 getenv() walks of the environment --
// and again only when rcx_reload_options() is called (tests and A/B tools flip a switch and reload).
#pragma once

namespace rcx {
namespace opt {

enum Id {
    FORCE_SPLIT, FORCE_GENERIC, LANES, LANES_WAVES, LANES_NI, WGRAD_CPL, CPT, CPT_GRID, CPT_MX, CPL, CPL7, CPL14, CPL14_LDS, CPL14_MX, UPADD_CPL,
    ATTN_MFMA, ATTN_SCALAR, TRAIN_FUSED, BWD_SPLIT, BWD_NESTED, BWD_FUSED, PLANE_LPP, PLANE_B2, PLANE_NT, PLANE_ABLATE, LANES_ABLATE, COUNT
};

// the variable's value as of the last (re)load, or nullptr when it is not set; the pointer stays valid until the next reload
const char* value(Id id);
inline bool is_zero(Id id) { const char* v = value(id); return v && *v == '0'; }      // "RCX_X=0": switched off
inline bool is_on(Id id) { const char* v = value(id); return v && *v && *v != '0'; }  // set to anything but "" / "0..."
void reload();

}  // namespace opt
}  // namespace rcx
