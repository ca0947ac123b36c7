"""ctypes binding of libclsimhip.so (include/clsimhip.h).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing this module raises at import time of the symbols, and every
compute call fails loudly when no GPU is present (status CLSIMHIP_ERR_DEVICE).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLSIMHIP_LIB", os.path.join(HERE, "libclsimhip.so"))

OK, ERR_ARGUMENT, ERR_STATE, ERR_CONFIG, ERR_DEVICE, ERR_IO = 0, -1, -2, -3, -4, -5
REFINDEX_ICECUBE, REFINDEX_TABLE, REFINDEX_DISPERSION = 0, 1, 2       # include/clsimhip.h: CLSIMHIP_REFINDEX_*

DP = C.POINTER(C.c_double)


class Function(C.Structure):        # clsimhip_function
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("start", C.c_double), ("step", C.c_double),
                ("values", DP), ("value", C.c_double), ("wavelengths", DP)]


class RandomValue(C.Structure):     # clsimhip_random_value
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("first", C.c_double), ("spacing", C.c_double),
                ("y", DP), ("value", C.c_double), ("x", DP)]


class StepRequest(C.Structure):     # clsimhip_step_request
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("time", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float), ("length", C.c_float),
                ("pa", C.c_float), ("pb", C.c_float), ("kind", C.c_uint32), ("identifier", C.c_uint32),
                ("photons_per_step", C.c_uint32), ("num_photons_in_last_step", C.c_uint32), ("num_steps", C.c_uint64)]


class PPCConfig(C.Structure):       # clsimhip_ppc_config
    _fields_ = [("photons_per_step", C.c_uint32), ("high_photons_per_step", C.c_uint32), ("use_high_photons_per_step_from", C.c_double),
                ("use_cascade_extension", C.c_int32), ("reserved", C.c_int32), ("medium_density", C.c_double), ("seed", C.c_uint64)]


class Distribution(C.Structure):    # clsimhip_distribution
    _fields_ = [("kind", C.c_int32), ("value", C.c_float)]


class FlasherConfig(C.Structure):   # clsimhip_flasher_config
    _fields_ = [("polar", Distribution), ("azimuthal", Distribution), ("time_delay", Distribution),
                ("interpret_in_polar_coordinates", C.c_int32), ("photons_per_step", C.c_uint32),
                ("max_bunch_size", C.c_uint32), ("bunch_size_granularity", C.c_uint32)]


class Axis(C.Structure):            # clsimhip_axis
    _fields_ = [("kind", C.c_int32), ("min", C.c_double), ("max", C.c_double), ("n_bins", C.c_uint32), ("power", C.c_uint32)]


class Polynomial(C.Structure):      # clsimhip_polynomial
    _fields_ = [("n", C.c_int32), ("coefficients", DP), ("range_min", C.c_double), ("range_max", C.c_double),
                ("underflow", C.c_double), ("overflow", C.c_double)]


class MediumDesc(C.Structure):      # clsimhip_medium_desc
    _fields_ = [("num_layers", C.c_int32), ("layers_z_start", C.c_double), ("layers_height", C.c_double),
                ("min_wavelength", C.c_double), ("max_wavelength", C.c_double),
                ("lengths_kind", C.c_int32), ("abs_length", DP), ("sca_length", DP),
                ("alpha", C.c_double), ("kappa", C.c_double), ("A", C.c_double), ("B", C.c_double),
                ("D", C.c_double), ("E", C.c_double),
                ("a_dust400", DP), ("delta_tau", DP), ("b400", DP),
                ("n", C.c_double * 5), ("g", C.c_double * 5),
                ("scatter_kind", C.c_int32), ("liu_fraction", C.c_double), ("mean_cosine", C.c_double),
                ("has_anisotropy", C.c_int32), ("aniso_azimuth", C.c_double), ("aniso_k1", C.c_double),
                ("aniso_k2", C.c_double),
                ("has_pre_transform", C.c_int32), ("pre_renormalize", C.c_int32), ("pre_matrix", C.c_double * 9),
                ("has_post_transform", C.c_int32), ("post_renormalize", C.c_int32), ("post_matrix", C.c_double * 9),
                ("has_tilt", C.c_int32), ("tilt_num_distances", C.c_int32), ("tilt_num_z", C.c_int32),
                ("tilt_distances", DP), ("tilt_z_coordinates", DP), ("tilt_z_corrections", DP),
                ("tilt_azimuth", C.c_double),
                ("table_num_wavelengths", C.c_int32), ("table_start_wavelength", C.c_double),
                ("table_wavelength_step", C.c_double), ("table_store_as_16bit", C.c_int32),
                ("abs_length_table", DP), ("sca_length_table", DP),
                ("phase_index_kind", C.c_int32), ("group_index_kind", C.c_int32),
                ("phase_index_table", Function), ("group_index_table", Function)]


# every symbol include/clsimhip.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "clsimhip_medium_create", "clsimhip_medium_create_from_ppc",
    "clsimhip_medium_create_from_photonics", "clsimhip_medium_describe", "clsimhip_medium_destroy",
    "clsimhip_icecube_dom_acceptance", "clsimhip_make_cherenkov_wlen_generator", "clsimhip_make_wlen_generator",
    "clsimhip_mwc_multipliers", "clsimhip_mwc_multipliers_from_file", "clsimhip_seed_streams",
    "clsimhip_create", "clsimhip_destroy", "clsimhip_last_error", "clsimhip_set_device", "clsimhip_get_device", "clsimhip_uses_pooled_kernel", "clsimhip_kernel_for_bunch",
    "clsimhip_step_series_blob_size", "clsimhip_encode_step_series", "clsimhip_decode_step_series",
    "clsimhip_photon_series_blob_size", "clsimhip_encode_photon_series", "clsimhip_decode_photon_series", "clsimhip_encode_portable_uint",
    "clsimhip_comm_get_unique_id", "clsimhip_comm_create", "clsimhip_comm_destroy", "clsimhip_gather_hits",
    "clsimhip_comm_info", "clsimhip_comm_statistics",
    "clsimhip_set_wlen_generators", "clsimhip_set_wlen_bias", "clsimhip_set_medium_properties", "clsimhip_set_geometry",
    "clsimhip_set_geometry_from_text_file",
    "clsimhip_set_enable_double_buffering", "clsimhip_set_double_precision", "clsimhip_set_stop_detected_photons",
    "clsimhip_set_save_all_photons", "clsimhip_set_save_all_photons_prescale",
    "clsimhip_set_fixed_number_of_absorption_lengths", "clsimhip_set_dom_pancake_factor",
    "clsimhip_set_photon_history_entries", "clsimhip_set_workgroup_size", "clsimhip_set_max_num_workitems",
    "clsimhip_compile", "clsimhip_get_max_workgroup_size", "clsimhip_initialize", "clsimhip_initialize_with_streams",
    "clsimhip_is_initialized", "clsimhip_enqueue_steps", "clsimhip_get_conversion_result", "clsimhip_get_result_histories",
    "clsimhip_release_result",
    "clsimhip_get_workgroup_size", "clsimhip_get_max_num_workitems", "clsimhip_queue_size",
    "clsimhip_more_photons_available", "clsimhip_get_statistics", "clsimhip_propagate_device",
    "clsimhip_set_concurrent_device_launches",
    "clsimhip_replace_indices_with_ids", "clsimhip_kernel_time_ms", "clsimhip_get_table", "clsimhip_get_rng_state",
    "clsimhip_eval_math", "clsimhip_check_math_exhaustive", "clsimhip_version",
    "clsimhip_eval_device_function", "clsimhip_eval_device_random", "clsimhip_get_option",
    "clsimhip_set_tuning", "clsimhip_get_tuning", "clsimhip_tabulator_set_tuning",
    "clsimhip_count_generated_steps", "clsimhip_generate_steps_device", "clsimhip_generate_steps",
    "clsimhip_ppc_create", "clsimhip_ppc_destroy", "clsimhip_ppc_photons_per_meter", "clsimhip_ppc_enqueue", "clsimhip_shower_parameters",
    "clsimhip_flasher_correction_factor", "clsimhip_flasher_enqueue",
    "clsimhip_feeder_create", "clsimhip_feeder_destroy", "clsimhip_feeder_enqueue_light_source", "clsimhip_feeder_enqueue_steps",
    "clsimhip_feeder_enqueue_barrier", "clsimhip_feeder_barrier_active", "clsimhip_feeder_more_steps_available",
    "clsimhip_feeder_get_conversion_result", "clsimhip_feeder_release_result",
    "clsimhip_count_flasher_steps", "clsimhip_generate_flasher_steps_device", "clsimhip_generate_flasher_steps",
    "clsimhip_flasher_time_profile",
    "clsimhip_step_store_create", "clsimhip_step_store_destroy", "clsimhip_step_store_insert", "clsimhip_step_store_size",
    "clsimhip_step_store_count", "clsimhip_step_store_pop_bunch", "clsimhip_step_store_pop_bunch_filled",
    "clsimhip_step_store_size_with_dummy_fill",
    "clsimhip_tabulator_create", "clsimhip_tabulator_destroy", "clsimhip_tabulator_last_error",
    "clsimhip_tabulator_enqueue_steps", "clsimhip_tabulator_finish", "clsimhip_tabulator_get_shape",
    "clsimhip_tabulator_get_bin_content", "clsimhip_tabulator_get_bin_sums", "clsimhip_tabulator_get_bin_edges",
    "clsimhip_tabulator_get_statistics", "clsimhip_tabulator_get_rng_state", "clsimhip_tabulator_get_table",
    "clsimhip_tabulator_write_fits_file",
]

_lib = None


def load():
    """Loads libclsimhip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C clsim_amd/csrc); the HIP propagator has no fallback path" % LIB_PATH)
    # A process that also uses PyTorch must load torch's HIP runtime BEFORE this library pulls in /opt/rocm's: with the
    # order reversed torch finds "No HIP GPUs" (two copies of libamdhip64 with one SONAME).  The tests and bench.py use
    # torch for device buffers, so it is imported here when it is installed; the library itself does not need it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, sz, u32, u64, i32, dbl = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    sig = {
        "clsimhip_medium_create": (i32, [C.POINTER(MediumDesc), C.POINTER(vp)]),
        "clsimhip_medium_create_from_ppc": (i32, [C.c_char_p, dbl, i32, C.POINTER(vp)]),
        "clsimhip_medium_create_from_photonics": (i32, [C.c_char_p, dbl, C.POINTER(vp)]),
        "clsimhip_medium_describe": (i32, [vp, C.POINTER(MediumDesc)]),
        "clsimhip_medium_destroy": (None, [vp]),
        "clsimhip_icecube_dom_acceptance": (i32, [dbl, dbl, DP, DP, DP]),
        "clsimhip_make_cherenkov_wlen_generator": (i32, [C.POINTER(Function), vp, DP, DP, DP]),
        "clsimhip_make_wlen_generator": (i32, [C.POINTER(Function), C.POINTER(Function), vp, C.POINTER(RandomValue), DP, DP, C.c_size_t]),
        "clsimhip_mwc_multipliers": (i32, [vp, sz]),
        "clsimhip_mwc_multipliers_from_file": (i32, [C.c_char_p, vp, sz]),
        "clsimhip_seed_streams": (i32, [vp, sz, u64, vp]),
        "clsimhip_create": (i32, [i32, C.POINTER(vp)]),
        "clsimhip_destroy": (None, [vp]),
        "clsimhip_last_error": (C.c_char_p, [vp]),
        "clsimhip_set_device": (i32, [vp, i32]),
        "clsimhip_get_device": (i32, [vp, C.POINTER(i32)]),
        "clsimhip_uses_pooled_kernel": (i32, [vp, C.POINTER(i32)]),
        "clsimhip_kernel_for_bunch": (i32, [vp, sz, C.POINTER(i32)]),
        "clsimhip_step_series_blob_size": (i32, [sz, C.POINTER(sz)]),
        "clsimhip_encode_step_series": (i32, [vp, sz, vp, sz, C.POINTER(sz)]),
        "clsimhip_decode_step_series": (i32, [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "clsimhip_photon_series_blob_size": (i32, [sz, C.POINTER(sz)]),
        "clsimhip_encode_photon_series": (i32, [vp, sz, vp, sz, C.POINTER(sz)]),
        "clsimhip_decode_photon_series": (i32, [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "clsimhip_encode_portable_uint": (i32, [u64, vp, C.POINTER(sz)]),
        "clsimhip_comm_get_unique_id": (i32, [vp]),
        "clsimhip_comm_create": (i32, [i32, i32, i32, vp, C.POINTER(vp)]),
        "clsimhip_comm_destroy": (None, [vp]),
        "clsimhip_comm_info": (i32, [vp, vp, vp, vp, vp, C.c_size_t]),
        "clsimhip_comm_statistics": (i32, [vp, vp, vp, vp, vp, i32]),
        "clsimhip_gather_hits": (i32, [vp, vp, vp, sz, i32, vp, sz, vp, vp]),
        "clsimhip_set_wlen_generators": (i32, [vp, C.POINTER(RandomValue), sz]),
        "clsimhip_set_wlen_bias": (i32, [vp, C.POINTER(Function)]),
        "clsimhip_set_medium_properties": (i32, [vp, vp]),
        "clsimhip_set_geometry": (i32, [vp, sz, vp, vp, vp, vp, vp, C.POINTER(C.c_char_p), dbl]),
        "clsimhip_set_geometry_from_text_file": (i32, [vp, C.c_char_p, dbl, C.c_int32, C.c_int32, u32, u32]),
        "clsimhip_set_enable_double_buffering": (i32, [vp, i32]),
        "clsimhip_set_double_precision": (i32, [vp, i32]),
        "clsimhip_set_stop_detected_photons": (i32, [vp, i32]),
        "clsimhip_set_save_all_photons": (i32, [vp, i32]),
        "clsimhip_set_save_all_photons_prescale": (i32, [vp, dbl]),
        "clsimhip_set_fixed_number_of_absorption_lengths": (i32, [vp, dbl]),
        "clsimhip_set_dom_pancake_factor": (i32, [vp, dbl]),
        "clsimhip_set_photon_history_entries": (i32, [vp, u32]),
        "clsimhip_set_workgroup_size": (i32, [vp, sz]),
        "clsimhip_set_max_num_workitems": (i32, [vp, sz]),
        "clsimhip_compile": (i32, [vp]),
        "clsimhip_get_max_workgroup_size": (i32, [vp, C.POINTER(sz)]),
        "clsimhip_initialize": (i32, [vp, u64]),
        "clsimhip_initialize_with_streams": (i32, [vp, vp, vp, sz]),
        "clsimhip_is_initialized": (i32, [vp]),
        "clsimhip_enqueue_steps": (i32, [vp, vp, sz, u32]),
        "clsimhip_get_conversion_result": (i32, [vp, C.POINTER(u32), C.POINTER(vp), C.POINTER(sz)]),
        "clsimhip_get_result_histories": (i32, [vp, vp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(u32)]),
        "clsimhip_release_result": (i32, [vp, vp]),
        "clsimhip_get_workgroup_size": (i32, [vp, C.POINTER(sz)]),
        "clsimhip_get_max_num_workitems": (i32, [vp, C.POINTER(sz)]),
        "clsimhip_queue_size": (i32, [vp, C.POINTER(sz)]),
        "clsimhip_more_photons_available": (i32, [vp, C.POINTER(i32)]),
        "clsimhip_get_statistics": (i32, [vp, DP]),
        "clsimhip_propagate_device": (i32, [vp, vp, sz, sz, vp, sz, vp, vp]),
        "clsimhip_set_concurrent_device_launches": (i32, [vp, i32]),
        "clsimhip_replace_indices_with_ids": (i32, [vp, vp, sz]),
        "clsimhip_kernel_time_ms": (i32, [vp, i32, DP, C.POINTER(u64)]),
        "clsimhip_get_table": (C.c_long, [vp, C.c_char_p, DP, sz]),
        "clsimhip_get_rng_state": (i32, [vp, vp, sz]),
        "clsimhip_eval_math": (i32, [i32, i32, vp, vp, sz, vp]),
        "clsimhip_get_option": (i32, [vp, i32, DP]),
        "clsimhip_set_tuning": (i32, [vp, C.c_char_p, C.c_longlong]),
        "clsimhip_get_tuning": (i32, [vp, C.c_char_p, C.POINTER(C.c_longlong)]),
        "clsimhip_tabulator_set_tuning": (i32, [vp, C.c_char_p, C.c_longlong]),
        "clsimhip_eval_device_function": (i32, [vp, i32, i32, i32, vp, sz, vp]),
        "clsimhip_eval_device_random": (i32, [vp, i32, i32, i32, vp, vp, sz, sz, vp]),
        "clsimhip_check_math_exhaustive": (i32, [i32, i32, i32, i32, vp, sz]),
        "clsimhip_version": (C.c_char_p, []),
        "clsimhip_ppc_create": (i32, [vp, C.POINTER(Function), C.POINTER(PPCConfig), C.POINTER(vp)]),
        "clsimhip_ppc_destroy": (None, [vp]),
        "clsimhip_ppc_photons_per_meter": (i32, [vp, i32, C.POINTER(C.c_double)]),
        "clsimhip_ppc_enqueue": (i32, [vp, vp, sz, C.POINTER(StepRequest), sz, C.POINTER(sz)]),
        "clsimhip_shower_parameters": (i32, [i32, C.c_double, C.c_double, C.POINTER(C.c_double)]),
        "clsimhip_flasher_correction_factor": (i32, [C.POINTER(Function), C.c_double, C.POINTER(Function), C.c_double, C.c_double, C.POINTER(C.c_double)]),
        "clsimhip_flasher_enqueue": (i32, [C.c_double, u64, vp, sz, vp, sz, C.POINTER(sz)]),
        "clsimhip_feeder_create": (i32, [vp, i32, u64, sz, sz, sz, C.POINTER(vp)]),
        "clsimhip_feeder_destroy": (None, [vp]),
        "clsimhip_feeder_enqueue_light_source": (i32, [vp, vp]),
        "clsimhip_feeder_enqueue_steps": (i32, [vp, u32, vp, sz]),
        "clsimhip_feeder_enqueue_barrier": (i32, [vp]),
        "clsimhip_feeder_barrier_active": (i32, [vp, C.POINTER(C.c_int)]),
        "clsimhip_feeder_more_steps_available": (i32, [vp, C.POINTER(C.c_int)]),
        "clsimhip_feeder_get_conversion_result": (i32, [vp, C.c_double, C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_int)]),
        "clsimhip_feeder_release_result": (i32, [vp, vp]),
        "clsimhip_count_generated_steps": (i32, [C.POINTER(StepRequest), sz, sz, C.POINTER(sz), C.POINTER(sz)]),
        "clsimhip_generate_steps_device": (i32, [i32, C.POINTER(StepRequest), sz, u64, sz, vp, sz, vp, C.POINTER(sz)]),
        "clsimhip_generate_steps": (i32, [i32, C.POINTER(StepRequest), sz, u64, sz, vp, sz, C.POINTER(sz)]),
        "clsimhip_count_flasher_steps": (i32, [C.POINTER(FlasherConfig), vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "clsimhip_generate_flasher_steps_device": (i32, [i32, C.POINTER(FlasherConfig), vp, sz, u64, vp, sz, vp, C.POINTER(sz)]),
        "clsimhip_generate_flasher_steps": (i32, [i32, C.POINTER(FlasherConfig), vp, sz, u64, vp, sz, C.POINTER(sz)]),
        "clsimhip_flasher_time_profile": (i32, [dbl, vp, vp]),
        "clsimhip_step_store_create": (i32, [sz, C.POINTER(vp)]),
        "clsimhip_step_store_destroy": (None, [vp]),
        "clsimhip_step_store_insert": (i32, [vp, vp, sz]),
        "clsimhip_step_store_size": (i32, [vp, C.POINTER(sz)]),
        "clsimhip_step_store_count": (i32, [vp, u32, C.POINTER(u32)]),
        "clsimhip_step_store_pop_bunch": (i32, [vp, sz, vp, C.POINTER(sz)]),
        "clsimhip_step_store_pop_bunch_filled": (i32, [vp, sz, vp, vp]),
        "clsimhip_step_store_size_with_dummy_fill": (i32, [vp, sz, C.POINTER(sz)]),
        "clsimhip_tabulator_create": (i32, [i32, i32, C.POINTER(Axis), sz, i32, vp, C.POINTER(Function), C.POINTER(Polynomial),
                                            dbl, dbl, vp, vp, sz, C.POINTER(vp)]),
        "clsimhip_tabulator_destroy": (None, [vp]),
        "clsimhip_tabulator_last_error": (C.c_char_p, [vp]),
        "clsimhip_tabulator_enqueue_steps": (i32, [vp, vp, sz, DP]),
        "clsimhip_tabulator_finish": (i32, [vp]),
        "clsimhip_tabulator_get_shape": (i32, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "clsimhip_tabulator_get_bin_content": (i32, [vp, vp, sz, i32, i32]),
        "clsimhip_tabulator_get_bin_sums": (i32, [vp, vp, sz, i32]),
        "clsimhip_tabulator_get_bin_edges": (i32, [vp, i32, DP, sz]),
        "clsimhip_tabulator_get_statistics": (i32, [vp, DP]),
        "clsimhip_tabulator_get_rng_state": (i32, [vp, vp, sz]),
        "clsimhip_tabulator_get_table": (C.c_long, [vp, C.c_char_p, DP, sz]),
        "clsimhip_tabulator_write_fits_file": (i32, [vp, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), sz]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = sig[name]
    _lib = lib
    return lib
