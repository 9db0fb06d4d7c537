#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace google { struct NullStream { template <class T> NullStream &operator<<(const T &) { return *this; } NullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; } }; inline void InitGoogleLogging(const char *) {} }
#define CHECK(cond) if (cond) {} else ::google::NullStream()
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_NOTNULL(p) (p)
#define LOG(sev) ::google::NullStream()
#define LOG_IF(sev, c) ::google::NullStream()
#define VLOG(n) ::google::NullStream()
