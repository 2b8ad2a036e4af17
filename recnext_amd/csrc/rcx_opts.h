// The RCX_* environment switches (A/B measurement knobs; none changes results beyond the stated tolerances).  They are read ONCE, at the
// first call that asks for one, into process-wide storage -- a launch no longer pays ~35 getenv() walks of the environment --
// and again only when rcx_reload_options() is called (tests and A/B tools flip a switch and reload).  Header-only (inline functions with
// one shared table per process), so the stand-alone harnesses under tools/ that compile a single kernel file link without rcx_api.hip.
#pragma once
#include <stdlib.h>
#include <mutex>
#include <string>

namespace rcx {
namespace opt {

enum Id {
    FORCE_SPLIT, FORCE_GENERIC, LANES, LANES_WAVES, LANES_NI, WGRAD_CPL, CPT, CPT_GRID, CPL, CPL14, CPL14_LDS, CPL14_RL, UPADD_CPL, UPADD_CPT, NESTED,
    ATTN_MFMA, ATTN_SCALAR, ATTN_FUSED, TRAIN_FUSED, BWD_SPLIT, BWD_NESTED, BWD_FUSED, PLANE_LPP, PLANE_B2, PLANE_NT, PLANE_ABLATE, LANES_ABLATE, CPT_CB, CPT_STG, CPT16, MLP_STREAM, BWD_CPT, COUNT
};

namespace detail {
inline const char* name_of(int i)
{
    static const char* const k[COUNT] = {
        "RCX_FORCE_SPLIT", "RCX_FORCE_GENERIC", "RCX_LANES", "RCX_LANES_WAVES", "RCX_LANES_NI", "RCX_WGRAD_CPL", "RCX_CPT", "RCX_CPT_GRID",
        "RCX_CPL", "RCX_CPL14", "RCX_CPL14_LDS", "RCX_CPL14_RL", "RCX_UPADD_CPL", "RCX_UPADD_CPT", "RCX_NESTED", "RCX_ATTN_MFMA", "RCX_ATTN_SCALAR", "RCX_ATTN_FUSED", "RCX_TRAIN_FUSED",
        "RCX_BWD_SPLIT", "RCX_BWD_NESTED", "RCX_BWD_FUSED", "RCX_PLANE_LPP", "RCX_PLANE_B2", "RCX_PLANE_NT", "RCX_PLANE_ABLATE", "RCX_LANES_ABLATE",
        "RCX_CPT_CB", "RCX_CPT_STG", "RCX_CPT16", "RCX_MLP_STREAM", "RCX_BWD_CPT"};
    return k[i];
}
struct Table {
    std::string val[COUNT];
    bool set[COUNT];
    std::once_flag once;
    void read()
    {
        for (int i = 0; i < COUNT; ++i) {
            const char* v = getenv(name_of(i));
            set[i] = v != nullptr;
            val[i] = v ? v : "";
        }
    }
};
inline Table& table() { static Table t; return t; }
}  // namespace detail

// the variable's value as of the last (re)load, or nullptr when it is not set; the pointer stays valid until the next reload
inline const char* value(Id id)
{
    detail::Table& t = detail::table();
    std::call_once(t.once, [&] { t.read(); });
    return t.set[id] ? t.val[id].c_str() : nullptr;
}
inline bool is_zero(Id id) { const char* v = value(id); return v && *v == '0'; }      // "RCX_X=0": switched off
inline bool is_on(Id id) { const char* v = value(id); return v && *v && *v != '0'; }  // set to anything but "" / "0..."
inline void reload()
{
    detail::Table& t = detail::table();
    std::call_once(t.once, [&] { t.read(); });
    t.read();
}

}  // namespace opt
}  // namespace rcx
