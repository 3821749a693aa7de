// rtree.h -- RTree: the reference's insertion-built R-tree (RTree.cs:12-378) producing the
// bit-identical float array, built without the reference's redundant work.
//
// Same decisions, same order, same binary32 arithmetic as RTree.cs; what differs is HOW the
// state is kept ("exact-equivalent faster host build"):
//   * nodes live in one pool addressed by index, primitive boxes are computed once
//     (the C# getters allocate a new BoundingBox on every use, Sphere.cs:42-51,
//     Triangle.cs:64-80);
//   * a branch stores only children[0] and children[1].  RTree.cs:260-265 creates 7, but an
//     empty child has min=+FLT_MAX, max=-FLT_MAX (RTree.cs:334-339), so FullyContains is false
//     (BoundingBox.cs:28), Overlap is false (BoundingBox.cs:72), its GetSize() is -inf and its
//     score is 0*8 + v*2 + (v-(-inf))*10 = +inf, never < the sentinel (RTree.cs:226-239): it
//     can never receive a primitive, contributes the identity to UpdateBoundingBox
//     (RTree.cs:359-367) and is skipped by the serialiser (RTree.cs:153).  Children 2..6 are
//     therefore unobservable;
//   * UpdateBoundingBox (RTree.cs:332-377) re-derives every ancestor box up to the root after
//     each change AND again on every return of AddElement (RTree.cs:214, 250).  Boxes are pure
//     min/max functions of the leaves, so recomputing an unchanged node is a no-op: here a
//     change is propagated once and stops at the first ancestor whose box does not change.
// tests/test_rtree_parity.py checks the arrays bit-for-bit against the oracle's literal
// restatement.
#pragma once
#include <cstring>
#include <stdexcept>
#include <vector>

#include "scene_elements.h"

namespace rth {

class RTree {
  public:
    static constexpr int maximumChildNodes = 7;     // RTree.cs:18 (see header: only 2 are observable)
    static constexpr int maximumValuesPerNode = 5;  // RTree.cs:20

    // boxes: BoundingBox of primitive pointer i (spheres 0..S-1 then triangles, IScene.cs:27-35)
    explicit RTree(std::vector<BoundingBox> boxes) : primBoxes(std::move(boxes)) {
        nodes.reserve(primBoxes.size() + 16);
        rootNode = NewNode(-1);
    }

    // RTree.cs:49-52
    void AddPrimitive(int primitivePointer) { AddElement(rootNode, primitivePointer); }

    // RTree.cs:42-47 / 111-178
    std::vector<float> TurnIntoFloatArray() const {
        std::vector<float> out;
        if (nodes[rootNode].isEmpty) return out;
        out.reserve(nodes.size() * 10);
        Serialise(rootNode, out);
        return out;
    }

  private:
    struct TreeNode {
        bool isEmpty = true;
        bool isLeaf = true;
        int child[2] = {-1, -1};
        int count = 0;  // filled slots of primitivePointers (they fill front to back, RTree.cs:187-196)
        int primitivePointers[maximumValuesPerNode];
        BoundingBox boundingBox;
        int parent = -1;
    };
    std::vector<TreeNode> nodes;
    std::vector<BoundingBox> primBoxes;
    int rootNode;

    static BoundingBox EmptyBox() {
        const float mx = 3.40282347e+38f;
        return BoundingBox(Vector3(mx, mx, mx), Vector3(-mx, -mx, -mx));  // RTree.cs:334-339
    }
    int NewNode(int parent) {
        TreeNode n;
        n.parent = parent;
        n.boundingBox = EmptyBox();
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }

    BoundingBox Recompute(int ni) const {
        const TreeNode& n = nodes[ni];
        BoundingBox b = EmptyBox();
        auto grow = [&](const BoundingBox& c) {
            b.MinimumValues.X = MathF_Min(b.MinimumValues.X, c.MinimumValues.X);
            b.MinimumValues.Y = MathF_Min(b.MinimumValues.Y, c.MinimumValues.Y);
            b.MinimumValues.Z = MathF_Min(b.MinimumValues.Z, c.MinimumValues.Z);
            b.MaximumValues.X = MathF_Max(b.MaximumValues.X, c.MaximumValues.X);
            b.MaximumValues.Y = MathF_Max(b.MaximumValues.Y, c.MaximumValues.Y);
            b.MaximumValues.Z = MathF_Max(b.MaximumValues.Z, c.MaximumValues.Z);
        };
        if (n.isLeaf) {
            for (int i = 0; i < n.count; i++) grow(primBoxes[n.primitivePointers[i]]);
        } else {
            grow(nodes[n.child[0]].boundingBox);
            grow(nodes[n.child[1]].boundingBox);
        }
        return b;
    }
    static bool SameBits(const BoundingBox& a, const BoundingBox& b) {
        return std::memcmp(&a, &b, sizeof(BoundingBox)) == 0;
    }
    // RTree.cs:332-377, propagated once (see header)
    void UpdateBoundingBox(int ni) {
        while (ni >= 0) {
            BoundingBox b = Recompute(ni);
            if (SameBits(b, nodes[ni].boundingBox)) return;
            nodes[ni].boundingBox = b;
            ni = nodes[ni].parent;
        }
    }

    // RTree.cs:180-252
    void AddElement(int ni, int primitivePointer) {
        nodes[ni].isEmpty = false;
        if (nodes[ni].isLeaf) {
            if (nodes[ni].count < maximumValuesPerNode) {
                nodes[ni].primitivePointers[nodes[ni].count++] = primitivePointer;
                UpdateBoundingBox(ni);
                return;
            }
            SplitNode(ni);
            AddElement(ni, primitivePointer);
            return;
        }
        const BoundingBox bb = primBoxes[primitivePointer];
        for (int k = 0; k < 2; k++) {
            if (nodes[nodes[ni].child[k]].boundingBox.FullyContains(bb)) {
                AddElement(nodes[ni].child[k], primitivePointer);
                return;
            }
        }
        int index = -1;
        const float overlapWeight = 8.0f, sizeWeight = 2.0f, growthWeight = 10.0f;
        float lowestScore = 9999999999999.0f;
        for (int k = 0; k < 2; k++) {
            const BoundingBox& cb = nodes[nodes[ni].child[k]].boundingBox;
            float tempScore = 0;
            tempScore += cb.OverlapSize(bb) * overlapWeight;
            float tempSize = cb.CalculatePotentialBoundingBoxSize(bb);
            tempScore += tempSize * sizeWeight;
            tempScore += (tempSize - cb.GetSize()) * growthWeight;
            if (tempScore < lowestScore) {
                index = k;
                lowestScore = tempScore;
            }
        }
        if (index < 0)  // children[-1] in C#
            throw std::out_of_range("IndexOutOfRangeException: no child scored below the sentinel (RTree.cs:249)");
        AddElement(nodes[ni].child[index], primitivePointer);
    }

    // RTree.cs:256-330
    void SplitNode(int ni) {
        int ptrs[maximumValuesPerNode];
        for (int i = 0; i < maximumValuesPerNode; i++) ptrs[i] = nodes[ni].primitivePointers[i];
        int index0 = -1, index1 = -1;
        float distance = 0;
        for (int i = 0; i < maximumValuesPerNode; i++) {
            for (int j = 0; j < maximumValuesPerNode; j++) {
                if (i == j) continue;
                const BoundingBox &b0 = primBoxes[ptrs[i]], &b1 = primBoxes[ptrs[j]];
                float smallX = MathF_Min(b0.MinimumValues.X, b1.MinimumValues.X);
                float smallY = MathF_Min(b0.MinimumValues.Y, b1.MinimumValues.Y);
                float smallZ = MathF_Min(b0.MinimumValues.Z, b1.MinimumValues.Z);
                float bigX = MathF_Max(b0.MaximumValues.X, b1.MaximumValues.X);
                float bigY = MathF_Max(b0.MaximumValues.Y, b1.MaximumValues.Y);
                float bigZ = MathF_Max(b0.MaximumValues.Z, b1.MaximumValues.Z);
                float newDistance = (bigX - smallX) * (bigY - smallY) * (bigZ - smallZ);
                if (distance < newDistance) {
                    distance = newDistance;
                    index0 = i;
                    index1 = j;
                }
            }
        }
        if (index0 < 0)  // primitivePointers[-1] in C#
            throw std::out_of_range(
                "IndexOutOfRangeException: all pair volumes are 0 in a full leaf (RTree.cs:303)");
        const int c0 = NewNode(ni), c1 = NewNode(ni);  // may reallocate `nodes`: no references held
        nodes[ni].isLeaf = false;
        nodes[ni].child[0] = c0;
        nodes[ni].child[1] = c1;
        nodes[ni].count = 0;
        AddElement(c0, ptrs[index0]);
        AddElement(c1, ptrs[index1]);
        const int parent = nodes[ni].parent;
        for (int i = 0; i < maximumValuesPerNode; i++) {
            if (i == index0 || i == index1) continue;
            AddElement(parent >= 0 ? parent : ni, ptrs[i]);
        }
    }

    // RTree.cs:111-178: child offsets are relative to the node's own first float
    void Serialise(int ni, std::vector<float>& out) const {
        const TreeNode& n = nodes[ni];
        const size_t start = out.size();
        out.push_back(n.boundingBox.MinimumValues.X);
        out.push_back(n.boundingBox.MinimumValues.Y);
        out.push_back(n.boundingBox.MinimumValues.Z);
        out.push_back(n.boundingBox.MaximumValues.X);
        out.push_back(n.boundingBox.MaximumValues.Y);
        out.push_back(n.boundingBox.MaximumValues.Z);
        out.push_back(n.isLeaf ? 1.0f : 0.0f);
        if (n.isLeaf) {
            out.push_back((float)n.count);
            for (int i = 0; i < n.count; i++) out.push_back((float)n.primitivePointers[i]);
            return;
        }
        // both stored children are non-empty, so both have GetSize() >= 0 (RTree.cs:153)
        out.push_back(2.0f);
        const size_t slots = out.size();
        out.push_back(0);
        out.push_back(0);
        for (int k = 0; k < 2; k++) {
            out[slots + k] = (float)(out.size() - start);
            Serialise(n.child[k], out);
        }
    }
};

}  // namespace rth
