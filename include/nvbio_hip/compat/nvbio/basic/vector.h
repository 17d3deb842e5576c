// compat/nvbio/basic/vector.h -- nvbio::vector<system_tag,T> (nvbio/basic/vector.h:60-150): the reference's containers are thrust
// vectors selected by a system tag, with plain_view() / raw_pointer() to reach the storage.  rocThrust ships with ROCm, so the
// device flavour is thrust::device_vector (its begin() is what callers hand to FMIndexFilter::locate, batch_*_alignment_score and
// thrust algorithms); the host flavour is thrust::host_vector.  Needs a translation unit compiled by hipcc.
#pragma once
#include "types.h"
#include "vector_view.h"
#include <vector>
#if defined(__HIPCC__)
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

namespace nvbio {

template <typename system_tag, typename T> struct vector {};

template <typename T>
struct vector<host_tag, T> : public thrust::host_vector<T>
{
    typedef host_tag                                system_tag;
    typedef thrust::host_vector<T>                  base_type;
    typedef nvbio::vector_view<T*, uint64>          plain_view_type;
    typedef nvbio::vector_view<const T*, uint64>    const_plain_view_type;
    typedef plain_view_type                         device_view_type;
    operator plain_view_type()             { return plain_view_type(base_type::size(), base_type::empty() ? (T*)0 : thrust::raw_pointer_cast(&base_type::front())); }
    operator const_plain_view_type() const { return const_plain_view_type(base_type::size(), base_type::empty() ? (const T*)0 : thrust::raw_pointer_cast(&base_type::front())); }
    vector(const size_t size = 0, const T val = T()) : base_type(size, val) {}
    template <typename OtherVector> vector(const OtherVector& v) : base_type(v) {}
    template <typename OtherVector> vector& operator=(const OtherVector& v) { base_type::operator=(v); return *this; }
};
template <typename T>
struct vector<device_tag, T> : public thrust::device_vector<T>
{
    typedef device_tag                              system_tag;
    typedef thrust::device_vector<T>                base_type;
    typedef nvbio::vector_view<T*, uint64>          plain_view_type;
    typedef nvbio::vector_view<const T*, uint64>    const_plain_view_type;
    typedef plain_view_type                         device_view_type;
    operator plain_view_type()             { return plain_view_type(base_type::size(), base_type::empty() ? (T*)0 : thrust::raw_pointer_cast(&base_type::front())); }
    operator const_plain_view_type() const { return const_plain_view_type(base_type::size(), base_type::empty() ? (const T*)0 : thrust::raw_pointer_cast(&base_type::front())); }
    vector(const size_t size = 0, const T val = T()) : base_type(size, val) {}
    template <typename OtherVector> vector(const OtherVector& v) : base_type(v) {}
    template <typename OtherVector> vector& operator=(const OtherVector& v) { base_type::operator=(v); return *this; }
};

namespace cuda {
/// resize-and-copy between thrust vectors of either system (nvbio/basic/vector.h:44-69)
template <typename TTargetVector, typename TSourceVector>
inline void thrust_copy_vector(TTargetVector& target, TSourceVector& source)
{
    target.resize(source.size());
    thrust::copy(source.begin(), source.end(), target.begin());
}
template <typename TTargetVector, typename TSourceVector>
inline void thrust_copy_vector(TTargetVector& target, TSourceVector& source, uint32 count)
{
    target.resize(count);
    thrust::copy(source.begin(), source.begin() + count, target.begin());
}
} // namespace cuda

template <typename T> inline T*       raw_pointer(thrust::device_vector<T>& v)       { return v.empty() ? (T*)0 : thrust::raw_pointer_cast(&v.front()); }
template <typename T> inline const T* raw_pointer(const thrust::device_vector<T>& v) { return v.empty() ? (const T*)0 : thrust::raw_pointer_cast(&v.front()); }
template <typename T> inline T*       raw_pointer(thrust::host_vector<T>& v)         { return v.empty() ? (T*)0 : &v.front(); }
template <typename T> inline const T* raw_pointer(const thrust::host_vector<T>& v)   { return v.empty() ? (const T*)0 : &v.front(); }
/// plain views: sized views that decay to the raw pointer (vector.h:89-140, thrust_view.h:38-80)
template <typename T> struct device_view_subtype< thrust::device_vector<T> >      { typedef vector_view<T*, uint64> type; };
template <typename T> struct plain_view_subtype< thrust::host_vector<T> >         { typedef vector_view<T*, uint64> type; };
template <typename T> struct plain_view_subtype< thrust::device_vector<T> >       { typedef vector_view<T*, uint64> type; };
template <typename T> struct plain_view_subtype< const thrust::host_vector<T> >   { typedef vector_view<const T*, uint64> type; };
template <typename T> struct plain_view_subtype< const thrust::device_vector<T> > { typedef vector_view<const T*, uint64> type; };
template <typename T> inline vector_view<T*, uint64>       plain_view(thrust::device_vector<T>& v)        { return vector_view<T*, uint64>(v.size(), raw_pointer(v)); }
template <typename T> inline vector_view<const T*, uint64> plain_view(const thrust::device_vector<T>& v)  { return vector_view<const T*, uint64>(v.size(), raw_pointer(v)); }
template <typename T> inline vector_view<T*, uint64>       plain_view(thrust::host_vector<T>& v)          { return vector_view<T*, uint64>(v.size(), raw_pointer(v)); }
template <typename T> inline vector_view<const T*, uint64> plain_view(const thrust::host_vector<T>& v)    { return vector_view<const T*, uint64>(v.size(), raw_pointer(v)); }
template <typename T> inline vector_view<T*, uint64>       device_view(thrust::device_vector<T>& v)       { return vector_view<T*, uint64>(v.size(), raw_pointer(v)); }
template <typename T> inline vector_view<const T*, uint64> device_view(const thrust::device_vector<T>& v) { return vector_view<const T*, uint64>(v.size(), raw_pointer(v)); }
template <typename S, typename T> inline vector_view<T*, uint64>       plain_view(vector<S, T>& v)       { return plain_view(static_cast<typename vector<S, T>::base_type&>(v)); }
template <typename S, typename T> inline vector_view<const T*, uint64> plain_view(const vector<S, T>& v) { return plain_view(static_cast<const typename vector<S, T>::base_type&>(v)); }
template <typename T> inline typename thrust::device_vector<T>::iterator       begin(thrust::device_vector<T>& v)       { return v.begin(); }
template <typename T> inline typename thrust::device_vector<T>::const_iterator begin(const thrust::device_vector<T>& v) { return v.begin(); }
template <typename T> inline typename thrust::host_vector<T>::iterator         begin(thrust::host_vector<T>& v)         { return v.begin(); }
template <typename T> inline typename thrust::host_vector<T>::const_iterator   begin(const thrust::host_vector<T>& v)   { return v.begin(); }
template <typename T> struct plain_view_subtype< std::vector<T> >       { typedef vector_view<T*, uint64> type; };
template <typename T> struct plain_view_subtype< const std::vector<T> > { typedef vector_view<const T*, uint64> type; };
template <typename T> inline vector_view<T*, uint64>       plain_view(std::vector<T>& v)       { return vector_view<T*, uint64>(v.size(), v.empty() ? (T*)0 : v.data()); }
template <typename T> inline vector_view<const T*, uint64> plain_view(const std::vector<T>& v) { return vector_view<const T*, uint64>(v.size(), v.empty() ? (const T*)0 : v.data()); }
template <typename T> inline T*       raw_pointer(std::vector<T>& v)       { return v.empty() ? (T*)0 : v.data(); }
template <typename T> inline const T* raw_pointer(const std::vector<T>& v) { return v.empty() ? (const T*)0 : v.data(); }

namespace priv {
/// the raw address behind an iterator that is known to walk plain memory (NULL for anything else)
template <typename It> struct plain_iterator { static const bool ok = false; typedef void value_type; static void* get(It) { return 0; } };
template <typename T> struct plain_iterator<T*> { static const bool ok = true; typedef T value_type; static T* get(T* p) { return p; } };
template <typename T> struct plain_iterator< thrust::device_ptr<T> > { static const bool ok = true; typedef T value_type; static T* get(thrust::device_ptr<T> p) { return thrust::raw_pointer_cast(p); } };
template <typename T> struct plain_iterator< thrust::detail::normal_iterator< thrust::device_ptr<T> > > { static const bool ok = true; typedef T value_type;
    static T* get(thrust::detail::normal_iterator< thrust::device_ptr<T> > p) { return thrust::raw_pointer_cast(&*p); } };
template <typename T> struct plain_iterator< thrust::detail::normal_iterator<T*> > { static const bool ok = true; typedef T value_type;
    static T* get(thrust::detail::normal_iterator<T*> p) { return &*p; } };
} // namespace priv

} // namespace nvbio
#else
#include <vector>
namespace nvbio {
template <typename system_tag, typename T> struct vector {};
template <typename T> struct vector<host_tag, T> : public std::vector<T>
{
    typedef host_tag system_tag; typedef std::vector<T> base_type;
    vector(const size_t size = 0, const T val = T()) : base_type(size, val) {}
};
template <typename T> struct plain_view_subtype< std::vector<T> >       { typedef vector_view<T*, uint64> type; };
template <typename T> struct plain_view_subtype< const std::vector<T> > { typedef vector_view<const T*, uint64> type; };
template <typename T> inline vector_view<T*, uint64>       plain_view(std::vector<T>& v)       { return vector_view<T*, uint64>(v.size(), v.empty() ? (T*)0 : v.data()); }
template <typename T> inline vector_view<const T*, uint64> plain_view(const std::vector<T>& v) { return vector_view<const T*, uint64>(v.size(), v.empty() ? (const T*)0 : v.data()); }
template <typename T> inline T*       raw_pointer(std::vector<T>& v)       { return v.empty() ? (T*)0 : v.data(); }
template <typename T> inline const T* raw_pointer(const std::vector<T>& v) { return v.empty() ? (const T*)0 : v.data(); }
namespace priv {
template <typename It> struct plain_iterator { static const bool ok = false; typedef void value_type; static void* get(It) { return 0; } };
template <typename T> struct plain_iterator<T*> { static const bool ok = true; typedef T value_type; static T* get(T* p) { return p; } };
}
} // namespace nvbio
#endif
