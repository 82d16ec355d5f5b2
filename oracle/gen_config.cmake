# oracle/gen_config.cmake -- TEST INFRASTRUCTURE ONLY.
# Produces the reference's generated header config.h from the reference's OWN
# template (${REF}/config.h.in) with cmake's configure_file(), exactly the step
# the reference's top-level CMakeLists.txt performs, but in script mode so that
# the reference's build system (which hard-fails without ALSA,
# CMakeLists.txt:14-16) is never run.  No HAVE_* feature is defined: ALSA,
# MySQL, curl and PulseAudio development files are absent from this image, so
# the sinks compile to their own stubs (out_mysql.c:299-301, out_json.c:407-420).
# Usage: cmake -DREF=/root/reference -DOUT=<dir> -P gen_config.cmake
configure_file(${REF}/config.h.in ${OUT}/config.h)
