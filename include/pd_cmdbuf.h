/*
 * pd_cmdbuf.h — command buffers of libpd_hip.so (csrc/cmdbuf.hip): a recorded sequence of this library's C-ABI calls, replayed from
 * C++ by one call.  No reference counterpart: the reference issues its step through PyTorch eager mode (mask2former_transformer_decoder.py:
 * 380-447 and pixel_decoder/msdeformattn.py:96-175 are the loops whose launches this serves); this is the host-side mechanism that keeps
 * ~600 launches per step of the fused encoder / decoder cores from costing a Python call each.
 *
 * A command = index of a recordable function (pd_cmd_fn_index, every `int pd_*(...)` entry point that takes a stream) + its argument
 * words.  kind[i]:  PD_CMD_LITERAL  the word as recorded (integers sign-extended to 64 bits, float / double bit patterns, pointers
 *                                   into persistent memory);
 *                   PD_CMD_STREAM   replaced by the `stream` given to pd_cmd_replay;
 *                   s >= 0          the word is a BYTE OFFSET into slot s: replaced by slot_bases[s] + offset (the region's input
 *                                   tensors, whose addresses change from step to step).
 * pd_cmd_replay runs the commands in order on `stream` and stops at the first error.  Host memory a command points to (descriptor
 * tables of the grouped launches) must stay alive and unchanged for as long as the command buffer is used.
 */
#ifndef PD_CMDBUF_H
#define PD_CMDBUF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_CMD_MAX_ARGS 48
#define PD_CMD_LITERAL (-1)
#define PD_CMD_STREAM (-2)

typedef struct PdCmd {
  int32_t fn, nargs;
  uint64_t a[PD_CMD_MAX_ARGS];
  int16_t kind[PD_CMD_MAX_ARGS];
} PdCmd;

int pd_cmd_fn_index(const char *name);          /* -1: not recordable */
int pd_cmd_fn_nargs(int fn);
int pd_cmd_replay(const PdCmd *cmds, int count, const uint64_t *slot_bases, int nslots, void *stream);

/* hipMemsetAsync / device-to-device hipMemcpyAsync as recordable entry points (zero-fills and copies inside a recorded region) */
int pd_memset_async(void *dst, int value, int64_t bytes, void *stream);
int pd_memcpy_d2d_async(void *dst, const void *src, int64_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_CMDBUF_H */
