// exception.h -- error convention of the NiftyMatch boundary (reference: src/gpu/utils/exception.h:25-110).
// Same public names (Exception<Std>, RUNTIME_/LOGIC_/RANGE_EXCEPTION, handleException), independent implementation.
#ifndef _EXCEPTION_H_
#define _EXCEPTION_H_

#include <cstdlib>
#include <exception>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>

template <class Std_Exception>
class Exception : public Std_Exception {
public:
    // Always throws; the message carries file, line and a description like the reference's (exception.h:96-109).
    [[noreturn]] static void throw_it(const char *file, const int line, const char *detailed = "-")
    {
        std::ostringstream os;
        os << "Exception in file '" << file << "' in line " << line << "\n"
           << "Detailed description: " << detailed << "\n";
        throw Exception(os.str());
    }
    [[noreturn]] static void throw_it(const char *file, const int line, const std::string &detailed)
    {
        throw_it(file, line, detailed.c_str());
    }
    virtual ~Exception() throw() {}

private:
    Exception() : Std_Exception("Unknown Exception.\n") {}
    explicit Exception(const std::string &str) : Std_Exception(str) {}
};

template <class Exception_Typ>
inline void handleException(const Exception_Typ &ex)
{
    std::cerr << ex.what() << std::endl;
    exit(EXIT_FAILURE);
}

#define RUNTIME_EXCEPTION(msg) Exception<std::runtime_error>::throw_it(__FILE__, __LINE__, msg)
#define LOGIC_EXCEPTION(msg) Exception<std::logic_error>::throw_it(__FILE__, __LINE__, msg)
#define RANGE_EXCEPTION(msg) Exception<std::range_error>::throw_it(__FILE__, __LINE__, msg)

// Device-side failures: the reference prints file:line and exit()s inside the launcher (helper_cuda.h
// getLastCudaError / checkCudaErrors, e.g. kernels/convolution.cu:152,158). nm_check() restores that on top of the
// status codes of the C ABI.
extern "C" const char *nm_error_string(int status);
inline void nm_check_impl(int status, const char *what, const char *file, int line)
{
    if (status != 0) {
        std::cerr << file << "(" << line << ") : " << what << " : (" << status << ") " << nm_error_string(status)
                  << std::endl;
        exit(EXIT_FAILURE);
    }
}
#define nm_check(status, what) nm_check_impl((status), (what), __FILE__, __LINE__)

#endif
