// tick_inst.hip -- instantiates the tick kernels for ONE slot count (compile with -DRG_P=1..8).
#ifndef RG_P
#error "compile with -DRG_P=<slots>"
#endif
#define RG_TICK_INSTANTIATE
#include "rg_tick_kernels.h"

template void rg_launch_tick_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
template void rg_launch_tick_classes_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
template void rg_launch_tick_split_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, u64);
template void rg_launch_tick_list_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64,
                                          u64 *, const RgListOut &);
template void rg_launch_tick_fused_t<RG_P>(hipStream_t, const RgState &, const RgFused &, bool);
template void rg_launch_tick_send_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32,
                                          const RgSendCols &, bool);
template void rg_launch_flush_small_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *,
                                            const RgListOut &);
template void rg_launch_flush_small_send_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *,
                                                 const RgListOut &, const RgSmallSend &);
template void rg_launch_mailbox_t<RG_P>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *,
                                        const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
