// lcb_kernel_limits.h — constants shared by the device code (lcb_kernel.h) and the host code that lays out its tables (lcb_segments.h).
#ifndef LCB_KERNEL_LIMITS_H
#define LCB_KERNEL_LIMITS_H
// A position on the device is (segment, g): a segment is a run of whole chromosomes with fewer than 2^32 junction occurrences in total,
// g a 32-bit offset inside it. A "chromosome word" cw = (segment << LCB_SEG_SHIFT) | chromosome travels with every occurrence record
// and every path instance.
#define LCB_SEG_SHIFT 24u
#define LCB_CHR_MASK 0x00FFFFFFu
#define LCB_MAX_SEG 32u
#endif
