// RtMi355x.cs -- P/Invoke binding of librt_mi355x.so for the Phlipsje/Raytracing host (SURVEY.md 8f-2).
// Drop this file into the reference project and apply the edits to RayTracer.cs listed in
// INTEGRATION.md.  It cannot be compiled in this repository's build image (no .NET SDK); the same
// entry points are exercised through the ctypes bindings in raytracing_amd/_native.py.
using System;
using System.Runtime.InteropServices;
using OpenTK.SceneElements;   // SphereStruct, PlaneStruct, TriangleStruct (PrimitiveStructsForBuffer.cs)

namespace OpenTK
{
    internal static class RtMi355x
    {
        const string Lib = "rt_mi355x";   // librt_mi355x.so next to the executable or on LD_LIBRARY_PATH

        [DllImport(Lib)] internal static extern int rt_create(int deviceId, out IntPtr ctx);
        // one context over several GPUs of the node: rt_upload_scene replicates the scene, rt_render splits the frame
        // into 8-row bands over the devices and gathers them on deviceIds[0] over RCCL; same pixels as one GPU
        [DllImport(Lib)] internal static extern int rt_create_multi([In] int[] deviceIds, int nDevices, out IntPtr ctx);
        [DllImport(Lib)] internal static extern int rt_multi_info(IntPtr ctx, out int nDevices, out int usesRccl);
        [DllImport(Lib)] internal static extern void rt_destroy(IntPtr ctx);
        [DllImport(Lib)] internal static extern IntPtr rt_last_error(IntPtr ctx);   // UTF-8, owned by ctx
        // tuning knobs, results never depend on them (include/rt_mi355x.h RT_OPTION_*):
        // 3 = traversal (0 auto), 4 = tight tree on/off, 5 = node count from which a scene counts as large
        [DllImport(Lib)] internal static extern int rt_set_option(IntPtr ctx, int option, int value);

        // replaces SendPrimitivesToShader's GL block, RayTracer.cs:1007-1045
        [DllImport(Lib)] internal static extern int rt_upload_scene(IntPtr ctx,
            [In] SphereStruct[] spheres, int nSpheres,
            [In] PlaneStruct[] planes, int nPlanes,
            [In] TriangleStruct[] triangles, int nTriangles,
            [In] float[] accel, long accelLen,
            [In] float[] lights, int lightsLen);

        // replaces GL.Uniform1 of the camera (RayTracer.cs:955) + GL.DrawArrays (RayTracer.cs:135)
        // device-resident outputs (IntPtr = device pointers); frames of one size, frameStride in pixels;
        // up to 8 frames share a launch.  For hosts that keep frames in HBM (video encode, multi-GPU assembly).
        [DllImport(Lib)] internal static extern int rt_render_frames_device(IntPtr ctx, [In] float[] cameras14xN, int nFrames,
            int maxBounces, int flags, int bandRows, int rank, int world, long frameStride, IntPtr dRgb, IntPtr dRgba8,
            IntPtr dPrimaryId, IntPtr dFinalId, IntPtr dBounces, IntPtr stream);

        // rt_render_device for hosts that must not allocate while rendering (HIP graph capture, serving loops):
        // the staged pipeline's path state lives in a fixed 96 MiB array inside the library; larger frames -> -6
        [DllImport(Lib)] internal static extern int rt_render_device_noalloc(IntPtr ctx, [In] float[] camera14,
            int maxBounces, int bandRows, int rank, int world, IntPtr dRgb, IntPtr dRgba8, IntPtr dPrimaryId,
            IntPtr dFinalId, IntPtr dBounces, IntPtr stream);

        [DllImport(Lib)] internal static extern int rt_render(IntPtr ctx, [In] float[] camera14, int maxBounces,
            [Out] float[] rgbOrNull, [Out] byte[] rgba8OrNull,
            [Out] int[] primaryIdOrNull, [Out] int[] finalIdOrNull, IntPtr statsOrNull);

        internal static void Check(IntPtr ctx, int rc)
        {
            if (rc != 0)
                throw new InvalidOperationException($"rt error {rc}: {Marshal.PtrToStringUTF8(rt_last_error(ctx))}");
        }
    }

    // Headless driver replacing OpenTKApp.Main (OpenTK.cs:245-251): builds the scene exactly as RayTracer's
    // constructor does (RayTracer.cs:31-38), renders one frame and writes it as a binary PPM, top row first.
    internal static class HeadlessMain
    {
        internal static void Render(Scenes.IScene scene, float[] cameraData, SphereStruct[] sphereData,
                                    PlaneStruct[] planeData, TriangleStruct[] triangleData, float[] lights, string path)
        {
            // all GPUs of the node: RtMi355x.rt_create_multi(new[] {0, 1, 2, 3, 4, 5, 6, 7}, 8, out ctx) -- nothing else changes
            RtMi355x.Check(IntPtr.Zero, RtMi355x.rt_create(0, out IntPtr ctx));
            try
            {
                RtMi355x.Check(ctx, RtMi355x.rt_upload_scene(ctx, sphereData, sphereData.Length, planeData,
                    planeData.Length, triangleData, triangleData.Length, scene.AccelerationStructureData,
                    scene.AccelerationStructureData.Length, lights, lights.Length));
                int w = (int)cameraData[12], h = (int)cameraData[13];
                var rgba8 = new byte[w * h * 4];
                RtMi355x.Check(ctx, RtMi355x.rt_render(ctx, cameraData, 10, null, rgba8, null, null, IntPtr.Zero));
                using var f = System.IO.File.Create(path);
                var header = System.Text.Encoding.ASCII.GetBytes($"P6\n{w} {h}\n255\n");
                f.Write(header, 0, header.Length);
                for (int y = h - 1; y >= 0; y--)          // row 0 of the library is the bottom row
                    for (int x = 0; x < w; x++)
                        f.Write(rgba8, (y * w + x) * 4, 3);
            }
            finally { RtMi355x.rt_destroy(ctx); }
        }
    }
}
