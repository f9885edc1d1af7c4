// Shared by the translation units of libffn_hip.so (not part of the C-ABI).
#ifndef FFN_INTERNAL_H_
#define FFN_INTERNAL_H_

// Stores a printf-formatted message for ffn_last_error() (thread local) and
// returns `code`.  Defined in ffn_hip.hip.
int ffn_set_error(int code, const char* fmt, ...)
    __attribute__((format(printf, 2, 3)));

#endif  // FFN_INTERNAL_H_
