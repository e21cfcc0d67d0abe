// c3d_host.h — error plumbing shared by the host translation units of libc3d.so
#pragma once
#include <string>
namespace c3d {
int fail(int code, const std::string& msg);
}
