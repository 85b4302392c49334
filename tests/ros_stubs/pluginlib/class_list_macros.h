#pragma once
#include <type_traits>
#define PLUGINLIB_EXPORT_CLASS(cls, base) static_assert(std::is_base_of<base, cls>::value && !std::is_abstract<cls>::value, "plugin class must derive from its base and be concrete");
