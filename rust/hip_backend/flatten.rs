//! Scene flattener for the HIP backend: `&dyn Hittable` -> `Vec<dray_sphere>` without touching the scene API.
//!
//! `Hittable` exposes only `hit` (src/raytracer/hittable.rs:25-28) and the fields of `Sphere` and of the
//! materials are private (sphere.rs:7-9, material.rs:35,61-62,85), but every scene type derives `Serialize`
//! and the two traits are `typetag` traits with the tag `"type"` (hittable.rs:24, material.rs:4).  So the world
//! is serialised once into a small tree by the `Serializer` below -- no serde_json, serde is the only
//! dependency -- and the tree is walked: `HittableList` in insertion order (the tie-break order of
//! hittable_list.rs:65-70), `Sphere` with its material flattened in, `BoundingBox` as a `dray_box` record.
//! Anything else -- a hittable or a material added later -- is `Unsupported`, and the caller falls back to the
//! CPU loop.
//!
//! Add to the reference as `src/raytracer/hip_flatten.rs` (module `raytracer::hip_flatten`, feature `hip`).
use core::fmt;

use dray_hip_sys::{
    dray_box, dray_sphere, DRAY_MAT_DEFAULT, DRAY_MAT_DIELECTRIC, DRAY_MAT_LAMBERTIAN, DRAY_MAT_METAL,
    DRAY_MAT_TRANSPARENT, DRAY_SHAPE_BOX,
};
use serde::ser::{self, Serialize};

/// What serialising a scene object produces: just enough of a document tree for the scene types.
#[derive(Debug, Clone, PartialEq)]
pub enum Node {
    Unit,
    Bool(bool),
    Num(f64),
    Str(String),
    Seq(Vec<Node>),
    Map(Vec<(String, Node)>),
}

impl Node {
    fn get(&self, key: &str) -> Result<&Node, FlattenError> {
        match self {
            Node::Map(fields) => fields
                .iter()
                .find(|(k, _)| k == key)
                .map(|(_, v)| v)
                .ok_or_else(|| FlattenError::Shape(format!("field `{key}` is missing"))),
            _ => Err(FlattenError::Shape(format!("expected an object with a field `{key}`"))),
        }
    }
    fn num(&self) -> Result<f64, FlattenError> {
        match self {
            Node::Num(x) => Ok(*x),
            _ => Err(FlattenError::Shape("expected a number".into())),
        }
    }
    fn seq(&self) -> Result<&[Node], FlattenError> {
        match self {
            Node::Seq(v) => Ok(v),
            _ => Err(FlattenError::Shape("expected a sequence".into())),
        }
    }
    fn tag(&self) -> Result<&str, FlattenError> {
        match self.get("type")? {
            Node::Str(s) => Ok(s),
            _ => Err(FlattenError::Shape("the `type` tag is not a string".into())),
        }
    }
    /// `Vec3 { coords: [f64; 3] }` (vec3.rs:7-10)
    fn vec3(&self) -> Result<[f64; 3], FlattenError> {
        let c = self.get("coords")?.seq()?;
        if c.len() != 3 {
            return Err(FlattenError::Shape("a Vec3 has three coordinates".into()));
        }
        Ok([c[0].num()?, c[1].num()?, c[2].num()?])
    }
}

#[derive(Debug)]
pub enum FlattenError {
    /// a hittable or material the GPU backend has no form for: fall back to the CPU loop
    Unsupported(String),
    /// the serialised form is not what the scene types of this crate produce
    Shape(String),
    /// raised by a `Serialize` implementation
    Custom(String),
}

impl fmt::Display for FlattenError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            FlattenError::Unsupported(s) => write!(f, "no GPU form for {s}"),
            FlattenError::Shape(s) => write!(f, "unexpected scene layout: {s}"),
            FlattenError::Custom(s) => write!(f, "{s}"),
        }
    }
}
impl std::error::Error for FlattenError {}
impl ser::Error for FlattenError {
    fn custom<T: fmt::Display>(msg: T) -> Self {
        FlattenError::Custom(msg.to_string())
    }
}

/// The flat list `dray_render` takes, in the order `HittableList::hit` visits the objects.
pub fn flatten<T: Serialize + ?Sized>(world: &T) -> Result<Vec<dray_sphere>, FlattenError> {
    let tree = world.serialize(TreeSerializer)?;
    let mut out = Vec::new();
    walk_hittable(&tree, &mut out)?;
    Ok(out)
}

fn walk_hittable(node: &Node, out: &mut Vec<dray_sphere>) -> Result<(), FlattenError> {
    match node.tag()? {
        "HittableList" => {
            for object in node.get("objects")?.seq()? {
                walk_hittable(object, out)?; // insertion order
            }
            Ok(())
        }
        "Sphere" => {
            let mut s = dray_sphere::default();
            s.center = node.get("center")?.vec3()?;
            s.radius = node.get("radius")?.num()?;
            let mat = node.get("mat")?;
            match mat.tag()? {
                "DefaultMaterial" => s.kind = DRAY_MAT_DEFAULT,
                "Transparent" => s.kind = DRAY_MAT_TRANSPARENT,
                "Lambertian" => {
                    s.kind = DRAY_MAT_LAMBERTIAN;
                    s.albedo = mat.get("albedo")?.vec3()?;
                }
                "Metal" => {
                    s.kind = DRAY_MAT_METAL;
                    s.albedo = mat.get("albedo")?.vec3()?;
                    s.fuzz_or_ior = mat.get("fuzz")?.num()?;
                }
                "Dialectric" => {
                    s.kind = DRAY_MAT_DIELECTRIC;
                    s.fuzz_or_ior = mat.get("refraction_index")?.num()?;
                }
                other => return Err(FlattenError::Unsupported(format!("material `{other}`"))),
            }
            out.push(s);
            Ok(())
        }
        "BoundingBox" => {
            let axes = node.get("axes")?.seq()?;
            if axes.len() != 3 {
                return Err(FlattenError::Shape("a BoundingBox has three axes".into()));
            }
            let mut b = dray_box::default();
            for (a, interval) in axes.iter().enumerate() {
                b.min[a] = interval.get("min")?.num()?;
                b.max[a] = interval.get("max")?.num()?;
            }
            b.shape = DRAY_SHAPE_BOX;
            // a dray_box is a list record of a dray_sphere's size with `shape` at the same place (dray.h)
            const _: () = assert!(core::mem::size_of::<dray_box>() == core::mem::size_of::<dray_sphere>());
            out.push(unsafe { core::mem::transmute::<dray_box, dray_sphere>(b) });
            Ok(())
        }
        other => Err(FlattenError::Unsupported(format!("hittable `{other}`"))),
    }
}

// ---- the serializer -----------------------------------------------------------------------------------------

struct TreeSerializer;

pub struct SeqBuilder(Vec<Node>);
pub struct MapBuilder {
    fields: Vec<(String, Node)>,
    pending_key: Option<String>,
}

fn unsupported<T>(what: &str) -> Result<T, FlattenError> {
    Err(FlattenError::Shape(format!("{what} does not occur in a scene")))
}

impl ser::Serializer for TreeSerializer {
    type Ok = Node;
    type Error = FlattenError;
    type SerializeSeq = SeqBuilder;
    type SerializeTuple = SeqBuilder;
    type SerializeTupleStruct = SeqBuilder;
    type SerializeTupleVariant = ser::Impossible<Node, FlattenError>;
    type SerializeMap = MapBuilder;
    type SerializeStruct = MapBuilder;
    type SerializeStructVariant = ser::Impossible<Node, FlattenError>;

    fn serialize_bool(self, v: bool) -> Result<Node, FlattenError> { Ok(Node::Bool(v)) }
    fn serialize_i8(self, v: i8) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_i16(self, v: i16) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_i32(self, v: i32) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_i64(self, v: i64) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_u8(self, v: u8) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_u16(self, v: u16) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_u32(self, v: u32) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_u64(self, v: u64) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_f32(self, v: f32) -> Result<Node, FlattenError> { Ok(Node::Num(v as f64)) }
    fn serialize_f64(self, v: f64) -> Result<Node, FlattenError> { Ok(Node::Num(v)) } // infinities and NaN as they are
    fn serialize_char(self, v: char) -> Result<Node, FlattenError> { Ok(Node::Str(v.to_string())) }
    fn serialize_str(self, v: &str) -> Result<Node, FlattenError> { Ok(Node::Str(v.to_owned())) }
    fn serialize_bytes(self, _v: &[u8]) -> Result<Node, FlattenError> { unsupported("a byte string") }
    fn serialize_none(self) -> Result<Node, FlattenError> { Ok(Node::Unit) }
    fn serialize_some<T: Serialize + ?Sized>(self, value: &T) -> Result<Node, FlattenError> { value.serialize(self) }
    fn serialize_unit(self) -> Result<Node, FlattenError> { Ok(Node::Unit) }
    fn serialize_unit_struct(self, _name: &'static str) -> Result<Node, FlattenError> { Ok(Node::Unit) }
    fn serialize_unit_variant(self, _name: &'static str, _index: u32, variant: &'static str) -> Result<Node, FlattenError> {
        Ok(Node::Str(variant.to_owned()))
    }
    fn serialize_newtype_struct<T: Serialize + ?Sized>(self, _name: &'static str, value: &T) -> Result<Node, FlattenError> {
        value.serialize(self)
    }
    fn serialize_newtype_variant<T: Serialize + ?Sized>(
        self, _name: &'static str, _index: u32, _variant: &'static str, _value: &T,
    ) -> Result<Node, FlattenError> {
        unsupported("an enum variant with data")
    }
    fn serialize_seq(self, len: Option<usize>) -> Result<SeqBuilder, FlattenError> {
        Ok(SeqBuilder(Vec::with_capacity(len.unwrap_or(0))))
    }
    fn serialize_tuple(self, len: usize) -> Result<SeqBuilder, FlattenError> { Ok(SeqBuilder(Vec::with_capacity(len))) }
    fn serialize_tuple_struct(self, _name: &'static str, len: usize) -> Result<SeqBuilder, FlattenError> {
        Ok(SeqBuilder(Vec::with_capacity(len)))
    }
    fn serialize_tuple_variant(
        self, _name: &'static str, _index: u32, _variant: &'static str, _len: usize,
    ) -> Result<Self::SerializeTupleVariant, FlattenError> {
        unsupported("a tuple variant")
    }
    fn serialize_map(self, len: Option<usize>) -> Result<MapBuilder, FlattenError> {
        Ok(MapBuilder { fields: Vec::with_capacity(len.unwrap_or(0)), pending_key: None })
    }
    // typetag's internally tagged form of `dyn Hittable` / `dyn Material` arrives here: the struct of the concrete
    // type with one more field in front, ("type", "<type name>")
    fn serialize_struct(self, _name: &'static str, len: usize) -> Result<MapBuilder, FlattenError> {
        Ok(MapBuilder { fields: Vec::with_capacity(len), pending_key: None })
    }
    fn serialize_struct_variant(
        self, _name: &'static str, _index: u32, _variant: &'static str, _len: usize,
    ) -> Result<Self::SerializeStructVariant, FlattenError> {
        unsupported("a struct variant")
    }
}

impl ser::SerializeSeq for SeqBuilder {
    type Ok = Node;
    type Error = FlattenError;
    fn serialize_element<T: Serialize + ?Sized>(&mut self, value: &T) -> Result<(), FlattenError> {
        self.0.push(value.serialize(TreeSerializer)?);
        Ok(())
    }
    fn end(self) -> Result<Node, FlattenError> { Ok(Node::Seq(self.0)) }
}
impl ser::SerializeTuple for SeqBuilder {
    type Ok = Node;
    type Error = FlattenError;
    fn serialize_element<T: Serialize + ?Sized>(&mut self, value: &T) -> Result<(), FlattenError> {
        ser::SerializeSeq::serialize_element(self, value)
    }
    fn end(self) -> Result<Node, FlattenError> { Ok(Node::Seq(self.0)) }
}
impl ser::SerializeTupleStruct for SeqBuilder {
    type Ok = Node;
    type Error = FlattenError;
    fn serialize_field<T: Serialize + ?Sized>(&mut self, value: &T) -> Result<(), FlattenError> {
        ser::SerializeSeq::serialize_element(self, value)
    }
    fn end(self) -> Result<Node, FlattenError> { Ok(Node::Seq(self.0)) }
}
impl ser::SerializeMap for MapBuilder {
    type Ok = Node;
    type Error = FlattenError;
    fn serialize_key<T: Serialize + ?Sized>(&mut self, key: &T) -> Result<(), FlattenError> {
        match key.serialize(TreeSerializer)? {
            Node::Str(s) => {
                self.pending_key = Some(s);
                Ok(())
            }
            _ => unsupported("a map key that is not a string"),
        }
    }
    fn serialize_value<T: Serialize + ?Sized>(&mut self, value: &T) -> Result<(), FlattenError> {
        let key = self.pending_key.take().ok_or_else(|| FlattenError::Shape("a map value without a key".into()))?;
        self.fields.push((key, value.serialize(TreeSerializer)?));
        Ok(())
    }
    fn end(self) -> Result<Node, FlattenError> { Ok(Node::Map(self.fields)) }
}
impl ser::SerializeStruct for MapBuilder {
    type Ok = Node;
    type Error = FlattenError;
    fn serialize_field<T: Serialize + ?Sized>(&mut self, key: &'static str, value: &T) -> Result<(), FlattenError> {
        self.fields.push((key.to_owned(), value.serialize(TreeSerializer)?));
        Ok(())
    }
    fn end(self) -> Result<Node, FlattenError> { Ok(Node::Map(self.fields)) }
}
