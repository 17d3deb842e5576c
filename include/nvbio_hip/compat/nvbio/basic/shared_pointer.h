// compat/nvbio/basic/shared_pointer.h -- SharedPointer<T> (nvbio/basic/shared_pointer.h): the reference carries its own
// reference-counted pointer; the standard one has the interface its callers use (construction from a raw pointer, get(),
// ->, *, ==, use_count, reset).
#pragma once
#include <memory>

namespace nvbio {

/// (the reference's second parameter names the counter type, e.g. SharedPointer<Impl, AtomicInt32> for pointers shared across
/// threads; std::shared_ptr's count is atomic already, so it is accepted and ignored)
template <typename T, typename CounterT = long>
struct SharedPointer : public std::shared_ptr<T>
{
    typedef std::shared_ptr<T> base_type;
    SharedPointer() {}
    template <typename U> explicit SharedPointer(U* p) : base_type(p) {}
    template <typename U, typename D> SharedPointer(U* p, D d) : base_type(p, d) {}
    template <typename U, typename C> SharedPointer(const SharedPointer<U, C>& o) : base_type(static_cast<const std::shared_ptr<U>&>(o)) {}
    template <typename U, typename C> SharedPointer& operator=(const SharedPointer<U, C>& o) { base_type::operator=(static_cast<const std::shared_ptr<U>&>(o)); return *this; }
    SharedPointer& operator=(T* p) { base_type::reset(p); return *this; }
};

} // namespace nvbio
