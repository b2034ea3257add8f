// Control functions of the LDS bank-conflict tracer (benchmarks/lds_conflicts.py).  TEST / ANALYSIS INFRASTRUCTURE ONLY: compiled into a
// library of its own with -DEMU_LDS_TRACE -g, never into tests/emu/libemu_kernels.so.
#ifdef EMU_LDS_TRACE
#include <dlfcn.h>
#include <gfx950_prims.h>

extern "C" {
void emu_lds_trace_enable(int on) { emu::lds_trace_on() = on != 0; if (on) emu::lds_recs().clear(); }
long emu_lds_trace_count() { return (long)emu::lds_recs().size(); }
void emu_lds_trace_copy(uint32_t* off, uint16_t* tid, uint16_t* kind, uint64_t* pc) {
    const auto& v = emu::lds_recs();
    for (size_t i = 0; i < v.size(); ++i) { off[i] = v[i].off; tid[i] = v[i].tid; kind[i] = v[i].kind; pc[i] = (uint64_t)v[i].pc; }
}
uint64_t emu_lds_trace_base() {          // load address of this library: pc - base = the offset llvm-symbolizer wants
    Dl_info info;
    return dladdr((void*)&emu_lds_trace_base, &info) ? (uint64_t)info.dli_fbase : 0;
}
}
#endif
