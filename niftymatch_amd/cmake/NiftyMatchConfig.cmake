# NiftyMatchConfig.cmake -- find_package(NiftyMatch CONFIG) for the MI355X drop-in.
# Same contract as the reference's src/cmake/NiftyMatchConfig.cmake:1-46: the file is installed INTO the include
# directory <prefix>/include/nm, and defines
#   NiftyMatch_INCLUDE_DIR   (located through macros.h)
#   NiftyMatch_gpuutils_LIB, NiftyMatch_kernels_LIB, NiftyMatch_sift_LIB   (<prefix>/lib/nm)
#   NiftyMatch_LIBS, NiftyMatch_FOUND
# Additional (new): NiftyMatch_HIP_LIBS = the HIP runtime the static libraries need at link time.
set(NiftyMatch_PATH_SUFFIX nm)

find_path(NiftyMatch_INCLUDE_DIR
    NAMES macros.h
    PATHS ${CMAKE_CURRENT_LIST_DIR}/../../include
    PATH_SUFFIXES ${NiftyMatch_PATH_SUFFIX})

foreach(_nm_mod gpuutils kernels sift)
    find_library(NiftyMatch_${_nm_mod}_LIB
        NAMES ${_nm_mod}
        PATHS ${CMAKE_CURRENT_LIST_DIR}/../../lib
        PATH_SUFFIXES ${NiftyMatch_PATH_SUFFIX})
endforeach()

find_library(NiftyMatch_HIP_LIBS NAMES amdhip64 PATHS /opt/rocm/lib ENV ROCM_PATH PATH_SUFFIXES lib)

# link order: sift -> kernels -> gpuutils (reference: src/gpu/sift/CMakeLists.txt:12)
set(NiftyMatch_LIBS
    ${NiftyMatch_sift_LIB}
    ${NiftyMatch_kernels_LIB}
    ${NiftyMatch_gpuutils_LIB}
    ${NiftyMatch_HIP_LIBS})

include(FindPackageHandleStandardArgs)
find_package_handle_standard_args(
    NiftyMatch DEFAULT_MSG
    NiftyMatch_LIBS NiftyMatch_INCLUDE_DIR)
