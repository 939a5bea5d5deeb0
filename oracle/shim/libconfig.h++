/* oracle/shim/libconfig.h++ -- TEST INFRASTRUCTURE ONLY.
 * libconfig++ is absent from this image.  The oracle never parses a config file (it fills
 * device_t/channel_t by hand the way src/config.cpp does), so this shim only has to let the
 * reference's translation units compile: every accessor aborts if it is ever reached. */
#ifndef ORACLE_SHIM_LIBCONFIG_HPP
#define ORACLE_SHIM_LIBCONFIG_HPP
#include <cstdlib>
#include <exception>
#include <string>
namespace libconfig {
class ConfigException : public std::exception {};
class FileIOException : public ConfigException {};
class SettingException : public ConfigException {
   public:
    const char* getPath() const { return ""; }
};
class SettingNotFoundException : public SettingException {};
class SettingTypeException : public SettingException {};
class ParseException : public ConfigException {
   public:
    int getLine() const { return 0; }
    const char* getError() const { return ""; }
    const char* getFile() const { return ""; }
};
class Setting {
   public:
    enum Type { TypeNone = 0, TypeInt, TypeInt64, TypeFloat, TypeString, TypeBoolean, TypeGroup, TypeArray, TypeList };
    bool exists(const char*) const { std::abort(); }
    bool exists(const std::string&) const { std::abort(); }
    int getLength() const { std::abort(); }
    Type getType() const { std::abort(); }
    const char* getName() const { std::abort(); }
    const char* c_str() const { std::abort(); }
    const char* getPath() const { std::abort(); }
    bool isNumber() const { std::abort(); }
    bool isRoot() const { std::abort(); }
    unsigned int getSourceLine() const { std::abort(); }
    Setting& operator[](const char*) const { std::abort(); }
    Setting& operator[](const std::string&) const { std::abort(); }
    Setting& operator[](int) const { std::abort(); }
    operator bool() const { std::abort(); }
    operator int() const { std::abort(); }
    operator unsigned int() const { std::abort(); }
    operator long() const { std::abort(); }
    operator long long() const { std::abort(); }
    operator float() const { std::abort(); }
    operator double() const { std::abort(); }
    operator const char*() const { std::abort(); }
    operator std::string() const { std::abort(); }
    template <class T>
    bool lookupValue(const char*, T&) const { std::abort(); }
};
class Config {
   public:
    void readFile(const char*) { std::abort(); }
    Setting& getRoot() const { std::abort(); }
    Setting& lookup(const char*) const { std::abort(); }
    Setting& lookup(const std::string&) const { std::abort(); }
};
}  // namespace libconfig
#endif
