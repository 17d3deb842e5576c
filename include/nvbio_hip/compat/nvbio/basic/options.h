// compat/nvbio/basic/options.h -- typed look-ups in a string -> string option map with a default and an optional alias
// (nvbio/basic/options.h:38-170); nvBowtie's parameter and scoring-scheme parsers are written on these.
#pragma once
#include "types.h"
#include "console.h"
#include <stdlib.h>
#include <string>

namespace nvbio {
namespace priv {
/// the value stored under `name`, else under `alias` (when given), else NULL
template <typename options_type>
inline const std::string* find_option(const options_type& options, const char* name, const char* alias = NULL)
{
    typename options_type::const_iterator it = options.find(std::string(name));
    if (it == options.end() && alias) it = options.find(std::string(alias));
    return it == options.end() ? (const std::string*)NULL : &it->second;
}
} // namespace priv

template <typename O> inline bool   bool_option(const O& o, const char* name, const bool val)                     { const std::string* s = priv::find_option(o, name);        return s ? atoi(s->c_str()) != 0 : val; }
template <typename O> inline bool   bool_option(const O& o, const char* name1, const char* name2, const bool val) { const std::string* s = priv::find_option(o, name1, name2); return s ? atoi(s->c_str()) != 0 : val; }
template <typename O> inline uint32 uint_option(const O& o, const char* name, const uint32 val)                     { const std::string* s = priv::find_option(o, name);        return s ? uint32(atoi(s->c_str())) : val; }
template <typename O> inline uint32 uint_option(const O& o, const char* name1, const char* name2, const uint32 val) { const std::string* s = priv::find_option(o, name1, name2); return s ? uint32(atoi(s->c_str())) : val; }
template <typename O> inline int32  int_option(const O& o, const char* name, const int32 val)                       { const std::string* s = priv::find_option(o, name);        return s ? atoi(s->c_str()) : val; }
template <typename O> inline int32  int_option(const O& o, const char* name1, const char* name2, const uint32 val)  { const std::string* s = priv::find_option(o, name1, name2); return s ? atoi(s->c_str()) : int32(val); }
template <typename O> inline int64  int64_option(const O& o, const char* name, const int64 val)                     { const std::string* s = priv::find_option(o, name);        return s ? int64(atoi(s->c_str())) : val; }
template <typename O> inline int64  int64_option(const O& o, const char* name1, const char* name2, const uint32 val){ const std::string* s = priv::find_option(o, name1, name2); return s ? int64(atoi(s->c_str())) : int64(val); }
template <typename O> inline float  float_option(const O& o, const char* name, const float val)                     { const std::string* s = priv::find_option(o, name);        return s ? float(atof(s->c_str())) : val; }
template <typename O> inline float  float_option(const O& o, const char* name1, const char* name2, const uint32 val){ const std::string* s = priv::find_option(o, name1, name2); return s ? float(atof(s->c_str())) : float(val); }
template <typename O> inline std::string string_option(const O& o, const char* name, const char* val)                     { const std::string* s = priv::find_option(o, name);        return s ? *s : std::string(val); }
template <typename O> inline std::string string_option(const O& o, const char* name1, const char* name2, const char* val) { const std::string* s = priv::find_option(o, name1, name2); return s ? *s : std::string(val); }

/// "a,b" -> int2
template <typename O>
inline int2 int2_option(const O& o, const char* name, const int2 val)
{
    const std::string* s = priv::find_option(o, name);
    if (!s) return val;
    const size_t comma = s->find(',');
    if (comma == std::string::npos) { log_warning(stderr, "int2_option() : parsing error, missing comma\n"); return val; }
    return make_int2(atoi(s->substr(0, comma).c_str()), atoi(s->substr(comma + 1).c_str()));
}

} // namespace nvbio
