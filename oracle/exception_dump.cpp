// exception_dump.cpp -- TEST INFRASTRUCTURE. Pins the error convention of the drop-in boundary ("Errors"):
// throws through the three convenience macros of the header chosen on the command line and prints, as JSON, what a client's
// catch block observes -- the what() text, which std exception type catches it, and that the default message is "-".
//   -DEXCEPTION_HEADER='"/root/reference/src/gpu/utils/exception.h"'   the REFERENCE's own header, compiled unmodified
//                                                                      (plain host C++: the second and last piece of the
//                                                                      reference's path that builds without nvcc)
//   -DEXCEPTION_HEADER='"../niftymatch_amd/nm/exception.h"'            the product's drop-in header
// The reference build's output is committed as tests/golden/exception_ref.json (a fixture: data, not source);
// tests/test_exception_pinned.py holds the product header against it. The file name inside the messages is made
// independent of where this file lies with #line, so both builds print the same text. No reference source is copied: this
// file only uses the public macro names (utils/exception.h:77-110).
#include EXCEPTION_HEADER

#include <cstdio>
#include <cstring>
#include <string>

static void json_string(const char *s)
{
    std::putchar('"');
    for (; *s; ++s) {
        if (*s == '\n') std::printf("\\n");
        else if (*s == '"' || *s == '\\') std::printf("\\%c", *s);
        else std::putchar(*s);
    }
    std::putchar('"');
}

template <class Std, class F>
static void record(const char *name, F thrower, bool last)
{
    std::printf(" {\"macro\": \"%s\", ", name);
    try {
        thrower();
        std::printf("\"thrown\": false}");
    } catch (const Std &e) {                 // the macro's own std type catches it (Exception<Std> derives from Std)
        std::printf("\"thrown\": true, \"caught_as_its_std_type\": true, \"what\": ");
        json_string(e.what());
        std::printf("}");
    } catch (const std::exception &e) {
        std::printf("\"thrown\": true, \"caught_as_its_std_type\": false, \"what\": ");
        json_string(e.what());
        std::printf("}");
    }
    std::printf("%s\n", last ? "" : ",");
}

int main()
{
    std::printf("[\n");
#line 100 "client.cpp"
    record<std::runtime_error>("RUNTIME_EXCEPTION", [] { RUNTIME_EXCEPTION("Pyramid depth must be positive"); }, false);
#line 200 "client.cpp"
    record<std::logic_error>("LOGIC_EXCEPTION", [] { LOGIC_EXCEPTION(std::string("assertion failed: a < b")); }, false);
#line 300 "client.cpp"
    record<std::range_error>("RANGE_EXCEPTION", [] { RANGE_EXCEPTION("index 7 out of range [0, 5)"); }, false);
#line 400 "client.cpp"
    record<std::runtime_error>("throw_it default", [] { Exception<std::runtime_error>::throw_it("other.cu", 12); }, false);
#line 500 "client.cpp"
    record<std::logic_error>("throw_it empty", [] { Exception<std::logic_error>::throw_it("", 0, ""); }, true);
    std::printf("]\n");
    return 0;
}
