# DPGOConfig.cmake -- lets the reference wrapper's `find_package(DPGO REQUIRED)` (CMakeLists.txt:6) and
# `target_link_libraries(... DPGO)` (CMakeLists.txt:151-154) resolve to the MI355X library unchanged:
#   catkin build -DDPGO_DIR=/opt/dpgo_ros_amd/cmake
# Provides the imported target `DPGO` (shared library libdpgo_hip.so + the header-only facade include/DPGO/*.h)
# and the classic variables DPGO_INCLUDE_DIRS / DPGO_LIBRARIES.
get_filename_component(_DPGO_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(DPGO_INCLUDE_DIRS "${_DPGO_ROOT}/include")
set(DPGO_LIBRARY "${_DPGO_ROOT}/dpgo_ros_amd/libdpgo_hip.so")
if(NOT EXISTS "${DPGO_LIBRARY}")
  set(DPGO_FOUND FALSE)
  set(DPGO_NOT_FOUND_MESSAGE "libdpgo_hip.so is not built: run `make -C ${_DPGO_ROOT}/dpgo_ros_amd/csrc`")
  return()
endif()
if(NOT TARGET DPGO)
  add_library(DPGO SHARED IMPORTED)
  set_target_properties(DPGO PROPERTIES
    IMPORTED_LOCATION "${DPGO_LIBRARY}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${DPGO_INCLUDE_DIRS}"
    INTERFACE_COMPILE_FEATURES cxx_std_17)
  find_package(Eigen3 QUIET NO_MODULE)   # with Eigen present DPGO::Matrix is Eigen::MatrixXd (include/DPGO/DPGO_types.h)
  if(TARGET Eigen3::Eigen)
    set_property(TARGET DPGO APPEND PROPERTY INTERFACE_LINK_LIBRARIES Eigen3::Eigen)
  endif()
  find_package(Threads QUIET)
  if(TARGET Threads::Threads)
    set_property(TARGET DPGO APPEND PROPERTY INTERFACE_LINK_LIBRARIES Threads::Threads)
  endif()
endif()
set(DPGO_LIBRARIES DPGO)
set(DPGO_FOUND TRUE)
