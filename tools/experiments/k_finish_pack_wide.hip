// k_finish_pack_wide.hip -- EXPERIMENT (round 5), kept for the record; not compiled into the library.
// Finish + pack of a frame that has the device to itself as ONE launch of wide workgroups (4 x 256 rays each: 512 workgroups
// for the 128 x 4096 raster instead of 2 048), with finish_pack_body's chained prefix.  Dropped into ls_project.hip next to
// k_finish_pack, launched from ls_trace.cpp's single-frame path (grid = ceil(ray blocks / 4)), it passed the parity suites
// (tests/test_gpu_parity.py, test_gpu_cull.py, test_gpu_dropin.py: 140 tests with the path forced, big-footprint queue
// included) and was SLOWER: one frame in flight 25.4 - 25.6 us per frame against 24.3 - 24.9 for k_project_finish + k_pack
// (same box, alternating processes, tools/exp_wide.sh).  With the 2 048-workgroup form (E7.3: 26.6 against 25.4) and the
// 256-workgroup form inside frame graphs (+ 1.0 us) that makes three sizes at which a kernel whose workgroups publish a word
// and read each other's through memory loses against a dependent launch (~4.5 us) on this device.
// (ls_project.hip's types and helpers -- ProjectParams, FinishPackArgs, BigItem, tri_test, kCullChunk -- are used as they are.)
// ------------------------------------------------------------------------------------------\n// A frame that has the device to itself")
b=found.index("