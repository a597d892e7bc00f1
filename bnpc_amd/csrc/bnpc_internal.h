// internal helpers shared by the translation units of libbnpc_hip.so
#ifndef BNPC_INTERNAL_H
#define BNPC_INTERNAL_H
#include <stdarg.h>

void bnpc_set_error(const char *fmt, ...);

#endif
