from .scene import Scene, Sphere, Quad, Mesh, Transform, Camera, ObjReader, build_bvh, uniforms_array  # noqa: F401
