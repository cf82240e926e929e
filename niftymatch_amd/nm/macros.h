// macros.h -- drop-in for NiftyMatch's src/utils/macros.h:1-8 (the file NiftyMatchConfig.cmake:16-19 searches for).
#ifndef __MACROS_H__
#define __MACROS_H__

#define DISALLOW_COPY_AND_ASSIGNMENT(TypeName) \
    TypeName(const TypeName &) = delete;       \
    void operator=(const TypeName &) = delete

#endif
