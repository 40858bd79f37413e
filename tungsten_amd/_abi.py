"""ctypes mirror of include/tungsten_amd.h (keep in sync; tests/test_abi.py checks sizes)."""
import ctypes as C

TGPU_ABI_VERSION = 4

f32 = C.c_float
i32 = C.c_int32
u32 = C.c_uint32
u64 = C.c_uint64
pf32 = C.POINTER(C.c_float)
pu32 = C.POINTER(C.c_uint32)
pi32 = C.POINTER(C.c_int32)
pu8 = C.POINTER(C.c_uint8)


class Texture(C.Structure):
    _fields_ = [("type", i32), ("value", f32*3), ("on_color", f32*3), ("off_color", f32*3),
                ("res_u", i32), ("res_v", i32), ("bitmap", i32)]


class Bitmap(C.Structure):
    _fields_ = [("w", i32), ("h", i32), ("clamp", i32), ("linear", i32), ("scale", f32), ("texels", pf32), ("average", f32*3)]


class Bsdf(C.Structure):
    _fields_ = [("type", i32), ("albedo_tex", i32), ("roughness_tex", i32), ("distribution", i32),
                ("eta", f32*3), ("k", f32*3), ("ior", f32), ("thickness", f32), ("sigma_a", f32*3),
                ("substrate", i32), ("scaled_sigma_a", f32*3), ("avg_transmittance", f32), ("enable_refraction", i32),
                ("diffuse_fresnel", f32), ("substrate_weight", f32), ("exponent", f32), ("diffuse_ratio", f32), ("aux_bsdf", i32), ("aux_tex", i32)]


class Medium(C.Structure):
    _fields_ = [("sigma_a", f32*3), ("sigma_s", f32*3), ("sigma_t", f32*3), ("absorption_only", i32), ("max_bounce", i32),
                ("transmittance", i32), ("param", f32*3), ("phase", i32), ("phase_g", f32),
                ("tr_a_type", i32), ("tr_a_param", f32*3), ("tr_b_type", i32), ("tr_b_param", f32*3), ("tr_ratio", f32)]


class Quad(C.Structure):
    _fields_ = [("base", f32*3), ("edge0", f32*3), ("edge1", f32*3), ("normal", f32*3), ("tangent", f32*3),
                ("bitangent", f32*3), ("inv_uv_sq", f32*2), ("area", f32), ("bsdf", i32), ("emission_tex", i32),
                ("int_medium", i32), ("ext_medium", i32)]


class Cube(C.Structure):
    _fields_ = [("rot", f32*9), ("inv_rot", f32*9), ("pos", f32*3), ("scale", f32*3), ("bsdf", i32),
                ("emission_tex", i32), ("int_medium", i32), ("ext_medium", i32)]


class Sphere(C.Structure):
    _fields_ = [("pos", f32*3), ("radius", f32), ("rot", f32*9), ("inv_rot", f32*9), ("bsdf", i32), ("emission_tex", i32),
                ("int_medium", i32), ("ext_medium", i32)]


class Disk(C.Structure):
    _fields_ = [("center", f32*3), ("r", f32), ("n", f32*3), ("tangent", f32*3), ("bitangent", f32*3), ("cos_apex", f32),
                ("cone_base", f32*3), ("bsdf", i32), ("emission_tex", i32), ("int_medium", i32), ("ext_medium", i32)]


class Env(C.Structure):
    _fields_ = [("present", i32), ("samplable", i32), ("emission_tex", i32), ("rot", f32*9), ("inv_rot", f32*9),
                ("dist_w", i32), ("dist_h", i32), ("pdf", pf32), ("cdf", pf32), ("marginal_pdf", pf32),
                ("marginal_cdf", pf32)]


class Camera(C.Structure):
    _fields_ = [("pos", f32*3), ("rot", f32*9), ("res_x", u32), ("res_y", u32), ("plane_dist", f32),
                ("ratio", f32), ("pixel_size_x", f32), ("filter_type", i32), ("filter_bin_size", f32),
                ("filter_cdf", f32*32), ("medium", i32)]


class Settings(C.Structure):
    _fields_ = [("min_bounces", i32), ("max_bounces", i32), ("enable_light_sampling", i32),
                ("enable_two_sided_shading", i32), ("enable_consistency_checks", i32), ("use_sobol", i32),
                ("adaptive_sampling", i32), ("enable_volume_light_sampling", i32)]


OUTPUT_COLOR, OUTPUT_DEPTH, OUTPUT_NORMAL, OUTPUT_ALBEDO, OUTPUT_VISIBILITY, OUTPUT_COUNT = range(6)
PART_A, PART_B, PART_VARIANCE, PART_COUNT = range(4)
OUTPUT_CHANNELS = (3, 1, 3, 3, 1)
OUTPUT_NAMES = ("color", "depth", "normal", "albedo", "visibility")


class OutputBuffer(C.Structure):
    _fields_ = [("enabled", i32), ("two_buffer_variance", i32), ("sample_variance", i32)]


class Light(C.Structure):
    _fields_ = [("kind", i32), ("index", i32)]


class Master(C.Structure):
    _fields_ = [("kind", i32), ("first", u32), ("count", u32), ("bounds_min", f32*3), ("bounds_max", f32*3)]


class Instances(C.Structure):
    _fields_ = [("n_masters", u32), ("masters", C.POINTER(Master)), ("n_instances", u32), ("pos", pf32), ("rot", pf32),
                ("master_id", pu8)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", u32), ("camera", Camera), ("settings", Settings), ("output_buffers", OutputBuffer*5),
                ("n_tris", u32), ("tri_pos", pf32), ("tri_normal", pf32), ("tri_uv", pf32), ("tri_bsdf", pi32),
                ("tri_smooth", pu8), ("tri_media", C.POINTER(C.c_int16)),
                ("n_meshes", u32), ("mesh_first_tri", pu32), ("mesh_bounds", pf32),
                ("n_quads", u32), ("quads", C.POINTER(Quad)),
                ("n_cubes", u32), ("cubes", C.POINTER(Cube)),
                ("n_spheres", u32), ("spheres", C.POINTER(Sphere)),
                ("n_disks", u32), ("disks", C.POINTER(Disk)),
                ("n_bsdfs", u32), ("bsdfs", C.POINTER(Bsdf)),
                ("n_textures", u32), ("textures", C.POINTER(Texture)),
                ("n_bitmaps", u32), ("bitmaps", C.POINTER(Bitmap)),
                ("n_lights", u32), ("lights", C.POINTER(Light)),
                ("n_media", u32), ("media", C.POINTER(Medium)),
                ("env", Env),
                ("n_master_tris", u32), ("n_master_quads", u32), ("n_master_cubes", u32),
                ("n_instance_sets", u32), ("instance_sets", C.POINTER(Instances)),
                ("sobol_matrices", pu32)]


class ShardDesc(C.Structure):
    _fields_ = [("n_pixels", u32), ("pixel_index", pu32), ("tile_seed", pu32)]


class Stats(C.Structure):
    _fields_ = [("samples", u64), ("extend_rays", u64), ("shadow_rays", u64),
                ("extend_node_visits", u64), ("extend_tri_tests", u64),
                ("shadow_node_visits", u64), ("shadow_tri_tests", u64),
                ("surface_events", u64),
                ("extend_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
                ("other_ms", C.c_double),
                ("extend_launches", u64), ("shadow_launches", u64), ("tail_launches", u64), ("shadow_vertices", u64),
                ("retire_ms", C.c_double), ("shade_launches", u64), ("retire_launches", u64),
                ("upload_ms", C.c_double), ("bvh_build_ms", C.c_double), ("bvh_on_device", u64)]


class SampleRecord(C.Structure):
    _fields_ = [("sample_count", u32), ("mean", f32), ("running_variance", f32)]
